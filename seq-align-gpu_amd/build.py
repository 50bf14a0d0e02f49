"""Build libswg.so (gfx950 kernels + C-ABI + plain-C host helpers) and the
smith_waterman CLI, in-tree, with explicit compiler invocations.

    python seq-align-gpu_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  The built files are git-ignored
but travel to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libswg.so")
CLI = os.path.join(HERE, "bin", "smith_waterman")

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

# The SI load/store optimizer would fuse the kernel's conflict-free ds_read_b64
# pairs into ds_read2_b64, which is banked modulo 32 dwords and runs at half the
# bytes per clock (MI355X_MICROARCH.md, LDS table): keep it off for device code.
# (The host half of the compile prints a harmless "not a recognized feature".)
#
# Machine scheduler: the AMDGPU back end's "max-ilp" strategy instead of its default (maximum
# occupancy).  The fill kernels are bounded by their launch bounds to the occupancy they run at anyway;
# within that budget the default strategy left the K=32 multi-pass kernel 14-19 registers short
# (spills inside the row loop), max-ilp fits it in 158 with no spill.  Measured on one box, same
# source: config 4 6933 -> 7178 GCUPS, config 5 6505 -> 6512, config 2 6284 -> 6283, config 3
# 7150 -> 7180 (profiles/r02_sched_strategy_ab.txt); round 3's kernels (f16 cells), alternating the two builds on
# one box: config 4 8240 -> 8460, config 5 7470 -> 7870, configs 2 and 3 level (profiles/r03_sched_strategy_ab.txt).
# The option is read by the AMDGPU target
# only; the host half of the compile ignores it.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt",
                "-mllvm", "-amdgpu-sched-strategy=max-ilp"]

HIP_SOURCES = ["swg_kernels.hip", "swg_trace.hip", "swg_api.cpp", "swg_group.cpp"]
KERNEL_PARTS = [0, 1, 2, 3, 4]  # swg_kernels.hip is compiled once per part (-DSWG_PART=n), in parallel
CXX_SOURCES = ["swg_pack.cpp", "swg_diag_host.cpp"]  # host-only C++, OpenMP via g++
C_SOURCES = ["swg_scoring.c", "swg_seqio.c", "swg_synth.c", "swg_threads.c"]
CLI_SOURCES = ["sw_cmdline.c"]


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    out = "\n".join(l for l in r.stdout.splitlines() if "not a recognized feature" not in l)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), out))
    if out.strip():
        print(out)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = []
    for d in (CSRC, HOST, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                hs.append(os.path.join(d, f))
    return hs


OBJDUMP = os.path.join(ROCM, "lib", "llvm", "bin", "llvm-objdump")
FAMILIES = ("swg_diag_dyn_kernel", "swg_diag_kernel", "swg_fill_kernelI8CellsI16", "swg_fill_kernelI9CellsSF16", "swg_diag32q_kernel", "swg_diag_qq_kernel")
ISA_STAMP = os.path.join(OBJ, "isa_checked")


def verify_isa(lib=LIB):
    """The int16 fill kernels must read their profile with ds_read_b64 and never with ds_read2_b64 (which is
    what DEVICE_FLAGS is for).  A compiler that drops or renames the feature would fuse the reads silently and
    halve the LDS rate: this looks at the built code object instead of trusting the flag.  Returns
    {kernel family: (ds_read_b64 count, ds_read2_b64 count)}; raises if a family has fused reads or none."""
    import glob
    import re
    import tempfile
    tmp = tempfile.mkdtemp()
    try:
        copy = os.path.join(tmp, "libswg.so")
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, cwd=tmp)
        counts = {}
        fam = None
        for co in sorted(glob.glob(copy + ".*gfx950*")):
            out = subprocess.run([OBJDUMP, "-d", co], stdout=subprocess.PIPE, text=True, check=True).stdout
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
                if m:
                    name = m.group(1)
                    fam = next((f for f in FAMILIES if f in name), None)
                    continue
                if fam and "ds_read" in line:
                    a, b = counts.get(fam, (0, 0))
                    counts[fam] = (a + ("ds_read_b64" in line), b + ("ds_read2_b64" in line))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for fam in FAMILIES:
        a, b = counts.get(fam, (0, 0))
        if a == 0 or b != 0:
            raise RuntimeError("%s: %d ds_read_b64 and %d ds_read2_b64 in the built kernels -- the load/store optimizer "
                               "switch (DEVICE_FLAGS) no longer keeps the profile reads unfused" % (fam, a, b))
    return counts


def _lib_digest():
    import hashlib
    h = hashlib.sha1()
    with open(LIB, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def ensure_isa_checked(verbose=True):
    """verify_isa() once per built library (a minute of disassembly): the stamp holds the library's digest,
    so a copy of the tree (the GPU box gets one) does not check again."""
    digest = _lib_digest()
    try:
        if open(ISA_STAMP).read().split()[0] == digest:
            return False
    except (OSError, IndexError):
        pass
    if not os.path.exists(OBJDUMP):
        # (the library linked fine; only the look at its disassembly is not possible here)
        print("[isa] warning: %s not found -- the ds_read2_b64 check of the built kernels was skipped" % OBJDUMP)
        return False
    counts = verify_isa()
    if verbose:
        print("[isa]", ", ".join("%s: %d ds_read_b64, no ds_read2_b64" % (k, v[0]) for k, v in sorted(counts.items())))
    os.makedirs(OBJ, exist_ok=True)
    open(ISA_STAMP, "w").write(digest + "\n" + repr(counts) + "\n")
    return True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    hdrs = _headers() + [os.path.abspath(__file__)]
    objs = []
    jobs = []  # (label, command) of everything that is out of date: run in parallel below
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        # the kernel file is compiled once per PART (see its header): its ~350 kernel instantiations as one translation
        # unit took four minutes, the parts side by side take as long as the largest of them
        parts = KERNEL_PARTS if src == "swg_kernels.hip" else [None]
        for part in parts:
            o = os.path.join(OBJ, src + (".o" if part is None else ".part%d.o" % part))
            if force or _stale(o, [s] + hdrs):
                jobs.append(("[hipcc] " + src + ("" if part is None else " part %d" % part),
                             [HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-c"] + DEVICE_FLAGS
                             + ([] if part is None else ["-DSWG_PART=%d" % part]) + ["-o", o, "-x", "hip", s]))
            objs.append(o)
    for src in CXX_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append(("[g++] " + src, ["g++", "-O2", "-fPIC", "-fopenmp", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                                          "-I" + os.path.join(ROCM, "include"), "-c", "-o", o, s]))
        objs.append(o)
    for src in C_SOURCES:
        s = os.path.join(HOST, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append(("[gcc] " + src, ["gcc", "-O2", "-fPIC", "-fopenmp", "-std=c11", "-Wall", "-c", "-o", o, s]))
        objs.append(o)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        if verbose:
            for label, _ in jobs:
                print(label)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 4)))) as pool:
            for _ in pool.map(lambda j: _run(j[1]), jobs):   # (the first failure is raised here)
                pass
    if force or _stale(LIB, objs):
        if verbose:
            print("[link] libswg.so")
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
             + ["-lgomp", "-lz", "-lm", "-ldl"])
    ensure_isa_checked(verbose)
    cli_srcs = [os.path.join(HOST, s) for s in CLI_SOURCES if os.path.exists(os.path.join(HOST, s))]
    if cli_srcs and (force or _stale(CLI, cli_srcs + [LIB] + hdrs)):
        if verbose:
            print("[gcc] smith_waterman")
        _run(["gcc", "-O2", "-std=c11", "-Wall", "-o", CLI] + cli_srcs
             + ["-L" + HERE, "-lswg", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(ROCM, "lib"),
                "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lz", "-lm"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("ok:", LIB)
