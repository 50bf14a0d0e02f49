"""Static instruction mix of one database row of the work-queue fill kernel, from the built library.

    python tools/row_isa.py [K] [edges] [wide|f16] > profiles/rNN_kK_row_isa.txt

Extracts the gfx950 code object from seq-align-gpu_amd/libswg.so (llvm-objdump --offloading),
disassembles swg_diag_dyn_kernel<K,16,false,false> and classifies the instructions of the first
unrolled row (from the end of the queue-event branch to the next one) -- what competes with the
recurrence for the VALU port and what issues beside it."""
import collections, glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 23
EDGES = int("edges" in sys.argv[2:])     # the several-pass variant (edge columns through HBM)
WIDE = 1 if "wide" in sys.argv[2:] else 2 if "f16" in sys.argv[2:] else 0   # the cells: 0 int16, 1 wide (to 65535), 2 packed f16 with max3
tmp = tempfile.mkdtemp()
lib = os.path.join(tmp, "libswg.so")
shutil.copy(os.path.join(ROOT, "seq-align-gpu_amd", "libswg.so"), lib)
subprocess.run([OBJDUMP, "--offloading", lib], stdout=subprocess.DEVNULL, check=True, cwd=tmp)
text = ""
for co in sorted(glob.glob(lib + ".*gfx950")):
    text += subprocess.run([OBJDUMP, "-d", co], stdout=subprocess.PIPE, text=True).stdout
lines = text.split("\n")
name = "_Z19swg_diag_dyn_kernelILi%dELi" % K     # <K, waves the instantiation was compiled for, no edges, not wide>
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <%s\d+ELb%dELi%dE" % (name, EDGES, WIDE), l))
end = start + 1
while end < len(lines) and not re.match(r"^[0-9a-f]+ <_Z", lines[end]):
    end += 1
body = lines[start:end]


def parse(l):
    m = re.match(r"^\s+(\S+)\s+(.*?)\s*//", l)
    return (m.group(1), m.group(2)) if m else None


def kind(op):
    if op.startswith("v_pk_") or op.startswith("v_perm"):
        return "VALU recurrence (v_pk_*, v_perm_b32)"
    if "dpp" in op:
        return "VALU DPP move (token / edge hand-over)"
    if op.startswith("v_"):
        return "VALU other (addresses, flags, selects)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "memory / other"


# rows are unrolled four times; a row's profile addresses are formed by two SDWA adds (residue bytes 0
# and 1 of the token): a row is taken from one BYTE_0 add to the next
starts = [i for i, l in enumerate(body) if ("v_add_u32_sdwa" in l or "v_xor_b32_sdwa" in l) and "BYTE_0" in l]
# the kernel holds its main loop twice (16-lane groups: row_shr hand-over; wider groups: wave_shr): take a
# row of the 16-lane loop, the one the headline shapes run
rows16 = [(x, y) for x, y in zip(starts, starts[1:]) if any("row_shr:1" in l for l in body[x:y]) and y - x < 700]
a, b = rows16[1]
mix = collections.Counter()
others = []
for l in body[a:b]:
    p = parse(l)
    if not p:
        continue
    k = kind(p[0] if "dpp" not in p[1] else p[0] + "_dpp")
    mix[k] += 1
    if not k.startswith("VALU recurrence"):
        others.append("    %-22s %s" % (p[0], p[1][:80]))
print("kernel %s...%s%s, %d instructions in all; one unrolled row = %d static instructions"
      % (name, " EDGES" * EDGES, " WIDE" * WIDE, len(body), b - a))
print("(a row of the loop for 16-lane groups; the skipped score-store / flag blocks are in the listing)")
for k, v in sorted(mix.items(), key=lambda kv: -kv[1]):
    print("  %4d  %s" % (v, k))
print("everything that is not the recurrence, in program order:")
print("\n".join(others))
shutil.rmtree(tmp)
