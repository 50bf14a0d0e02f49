cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/sweeps/cli_timing.py 100000 > gpurun_out/r2_cli_end_to_end.txt 2>&1; tail -60 gpurun_out/r2_cli_end_to_end.txt
# 64 queries of 128 aa against config 1's database through the tool (one pass)
python - <<'PY'
import os, sys, subprocess, importlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
swg = importlib.import_module("seq-align-gpu_amd")
L = "".join(chr(swg.lib.swg_index_letter(i)) for i in range(32))
def letters(idx):
    return idx.astype("uint8").tobytes().translate(bytes(ord(L[i]) if i < 32 else 63 for i in range(256)))
flat, off = swg.synth_db(0x5EED0001, 1024)
with open("/tmp/cli/db1.fa", "wb") as f:
    for i in range(1024):
        f.write(b">s%d\n" % i + letters(flat[int(off[i]):int(off[i + 1])]) + b"\n")
with open("/tmp/cli/q64.fa", "wb") as f:
    for i in range(65):
        f.write(b">q%d\n" % i + letters(swg.synth_query(100 + i, 128)) + b"\n")
cli = os.path.join(ROOT, "seq-align-gpu_amd", "bin", "smith_waterman")
mat = os.path.join(ROOT, "seq-align-gpu_amd", "data", "BLOSUM62.txt")
r = subprocess.run([cli, "--substitution_matrix", mat, "--timing", "--allqueries", "--files", "/tmp/cli/q64.fa", "/tmp/cli/db1.fa"],
                   stdout=open("/tmp/cli/out64.txt", "wb"), stderr=subprocess.PIPE, text=True)
print("== --allqueries, 65 queries of 128 aa vs 1024 sequences: exit", r.returncode)
print(r.stderr)
PY
