"""Build libhumanliff_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m humanliff_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhumanliff_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"] + os.environ.get("HL_FLAGS", "").split()   # HL_FLAGS: developer variants (-D...) for every file
# per-file overrides (the last -std wins): k_conv_wino4w's compile-time slot schedule uses templated lambdas
FILE_FLAGS = {"hl_conv_wino4w.hip": ["-std=c++20"] + os.environ.get("HL_W4W_FLAGS", "").split(), "hl_conv_h16.hip": ["-std=c++20"] + os.environ.get("HL_H16_FLAGS", "").split(),
              "hl_render.hip": os.environ.get("HL_RENDER_FLAGS", "").split()}   # HL_*_FLAGS: developer variants (-D...)


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    deps = _sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "humanliff_hip.h"))
    return deps


STAMP = LIB + ".flags"   # (next to the library: it travels with it)


def _flag_key():
    """Every flag the objects are compiled with (the HL_*_FLAGS developer variants included): a flags-only change must rebuild too -
    an A / B of two -D variants that silently compares a library with itself measures nothing."""
    return repr((HIPCC, FLAGS, sorted(FILE_FLAGS.items())))


def needs_build():
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(STAMP) or open(STAMP).read() != _flag_key():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library. Returns its path."""
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in _sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    audit_accumulator_file()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    with open(STAMP, "w") as f:
        f.write(_flag_key())
    return LIB


def audit_accumulator_file():
    """k_conv_wino4w keeps 16 accumulator tiles in the accumulator registers a[0:255] by NAME, inside inline asm (hl_conv_wino4w.hip).
    The compiler only knows them as clobbers; if register pressure ever makes it place a value of its own there (a load into a[..], a
    v_accvgpr_write, a spill) the kernel computes garbage without any diagnostic.  So the ISA of that file is checked after every build:
    outside the ;;#ASMSTART / ;;#ASMEND brackets no instruction may name an accumulator register."""
    import re
    import tempfile
    src = os.path.join(CSRC, "hl_conv_wino4w.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get("hl_conv_wino4w.hip", []) + ["--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "w4w.s")]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stdout.decode()}")
        inasm, bad = False, []
        for n, line in enumerate(open(os.path.join(tmp, "w4w.s")), 1):
            if "ASMSTART" in line:
                inasm = True
            elif "ASMEND" in line:
                inasm = False
            elif not inasm and not line.lstrip().startswith((";", ".")) and re.search(r"\ba\[?\d", line.split(";")[0]):
                bad.append(f"{n}: {line.strip()}")
    if bad:
        raise RuntimeError("k_conv_wino4w: the compiler touched the accumulator file outside the kernel's inline asm "
                           f"({len(bad)} instructions, e.g. {bad[:3]}): reduce the register pressure of the kernel")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
