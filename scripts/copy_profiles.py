"""Copy the summaries of scripts/refresh_profiles.sh (gpurun_out/refresh) into profiles/<round>_*.md, keeping the hand-written header
of each file (everything above its first table) and replacing the tables.   python scripts/copy_profiles.py [r02]"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P, rnd = os.path.join(ROOT, "gpurun_out", "refresh"), os.path.join(ROOT, "profiles"), (sys.argv[1] if len(sys.argv) > 1 else "r02")


def repl(name, src, marker):
    path = os.path.join(P, f"{rnd}_{name}")
    s = open(path).read()
    open(path, "w").write(s[:s.index(marker)] + open(os.path.join(R, src)).read())


shutil.copy(os.path.join(R, "bench.json"), os.path.join(P, f"{rnd}_bench.json"))
repl("bench_kernel_trace_single_stream.md", "trace_single.md", "| kernel | calls |")
repl("bench_kernel_trace_two_streams.md", "trace_overlap.md", "| kernel | calls |")
repl("pmc_hbm_traffic.md", "pmc_bench.md", "| kernels | launches |")
repl("pmc_sq_conv_kernels.md", "pmc_sq_conv.md", "| kernel | workgroups |")
repl("fit_kernel_trace.md", "trace_fit.md", "| kernel | calls |")
path = os.path.join(P, f"{rnd}_unet_train_kernel_trace.md")
s = open(path).read()
i = s.index("```\n"); j = s.index("```\n", i + 4); k = s.index("| kernel | calls |")
open(path, "w").write(s[:i + 4] + open(os.path.join(R, "unet_train_wall.txt")).read() + s[j:k] + open(os.path.join(R, "trace_unet_train.md")).read())
print("profiles updated from", R)
