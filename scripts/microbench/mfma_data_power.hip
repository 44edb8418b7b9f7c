// Does the rate of v_mfma_f32_32x32x16_f16 depend on the operand VALUES?  (It does - at the same reported shader clock and within 10 % of the same socket power: operands with more significant bits
// switch more of the multiplier array and the matrix unit issues fewer MFMAs per clock.)  Four waves per CU x 2 (one or two per SIMD), a long chain of independent MFMAs on register operands that are
//   zeros | fp16 subnormals with 3 significant bits (what the low plane of an unscaled |w| ~ 0.02 weight looks like) | normal fp16 with full mantissas (scaled planes)
// Prints TFLOP/s of each over ~0.3 s after a warm-up, with the shader clock (pp_dpm_sclk active level, hwmon freq1_input) and the socket power (hwmon power1_*) sampled every 20 ms meanwhile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <chrono>
#include <string>
#include <vector>
#include <fstream>
#include <glob.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const unsigned short *a, const unsigned short *b, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    h8 A[4], B[4];
    for (int i = 0; i < 4; ++i) { A[i] = *(const h8 *)(a + ((t * 4 + i) & 65535) * 8); B[i] = *(const h8 *)(b + ((t * 4 + i) & 65535) * 8); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[j], B[(i + j) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 1234.5f) out[0] = s;
}

// shader clock (active level of pp_dpm_sclk) and socket power (hwmon power1_average / power1_input, microwatts) sampled every 20 ms while a pattern runs
static std::vector<std::string> glob_paths(const char *pat) {
    std::vector<std::string> r;
    glob_t g;
    if (glob(pat, 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) r.push_back(g.gl_pathv[i]); globfree(&g); }
    return r;
}
struct Sampler {
    std::vector<std::string> fq = glob_paths("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input");
    double fq_sum = 0; int fq_n = 0;
    std::vector<std::string> clk = glob_paths("/sys/class/drm/card*/device/pp_dpm_sclk"), pw = glob_paths("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average");
    std::atomic<bool> stop{false};
    double clk_sum = 0, pw_sum = 0; int clk_n = 0, pw_n = 0;
    std::thread th;
    void start() {
        if (pw.empty()) pw = glob_paths("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input");
        stop = false; clk_sum = pw_sum = fq_sum = 0; clk_n = pw_n = fq_n = 0;
        th = std::thread([this] {
            while (!stop) {
                double cbest = 0, pbest = 0;
                for (auto &f : clk) { std::ifstream in(f); std::string l; while (std::getline(in, l)) if (!l.empty() && l.back() == '*') { double v = atof(l.c_str() + l.find(':') + 1); if (v > cbest) cbest = v; } }
                for (auto &f : pw) { std::ifstream in(f); double v = 0; if (in >> v) if (v > pbest) pbest = v; }
                double fbest = 0;
                for (auto &f : fq) { std::ifstream in(f); double v = 0; if (in >> v) if (v > fbest) fbest = v; }
                if (fbest > 0) { fq_sum += fbest * 1e-6; ++fq_n; }
                if (cbest > 0) { clk_sum += cbest; ++clk_n; }
                if (pbest > 0) { pw_sum += pbest * 1e-6; ++pw_n; }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    double fq_mhz = 0;
    void finish(double &mhz, double &watts) { stop = true; th.join(); mhz = clk_n ? clk_sum / clk_n : 0; watts = pw_n ? pw_sum / pw_n : 0; fq_mhz = fq_n ? fq_sum / fq_n : 0; }
};

int main() {
    const int n = 65536 * 8;
    unsigned short *h = new unsigned short[n];
    unsigned short *da, *db; float *dout;
    hipMalloc(&da, n * 2); hipMalloc(&db, n * 2); hipMalloc(&dout, 64);
    const char *names[] = {"zeros", "subnormal, 3 significant bits", "normal, 5 significant bits", "normal, full mantissa"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        srand(7);
        for (int i = 0; i < n; ++i) {
            unsigned short v = 0;
            const unsigned r = (unsigned)rand();
            if (mode == 1) v = (unsigned short)(((r & 7u) << 4) | ((r >> 8) & 1u) << 15);                          // exponent 0, mantissa bits 4..6
            if (mode == 2) v = (unsigned short)((((r >> 4) % 6 + 12) << 10) | ((r & 0xfu) << 6) | ((r >> 16) & 1u) << 15);
            if (mode == 3) v = (unsigned short)((((r >> 10) % 6 + 12) << 10) | (r & 0x3ffu) | ((r >> 16) & 1u) << 15);
            h[i] = v;
        }
        hipMemcpy(da, h, n * 2, hipMemcpyHostToDevice); hipMemcpy(db, h, n * 2, hipMemcpyHostToDevice);
        const int iters = 40000, grid = 512, reps = 16;
        k<<<grid, 256>>>(da, db, dout, iters);                  // warm-up at this operand pattern
        hipDeviceSynchronize();
        Sampler smp;
        smp.start();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) k<<<grid, 256>>>(da, db, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        double mhz, watts;
        smp.finish(mhz, watts);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)reps * grid * 4 * (double)iters * 16 * 2.0 * 32 * 32 * 16;
        printf("pass %d  %-32s %8.1f TFLOP/s (%.0f ms)   sclk level %.0f MHz, hwmon freq1 %.0f MHz   power %.0f W\n", rep, names[mode], flop / (ms * 1e-3) * 1e-12, ms, mhz, smp.fq_mhz, watts);
    }
    return 0;
}
