"""End-to-end wall time of the smith_waterman tool on a config-2-sized FASTA database (GPU box):
the phases the reference leaves out of its `Total Time` (reading, packing, upload, first search)."""
import importlib, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
swg = importlib.import_module("seq-align-gpu_amd")
L = "".join(chr(swg.lib.swg_index_letter(i)) for i in range(32))
def letters(idx):
    return idx.astype("uint8").tobytes().translate(bytes(ord(L[i]) if i < 32 else 63 for i in range(256)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
q = swg.synth_query(0x5EED0002, 367)
flat, off = swg.synth_db(0x5EED0002, n)
os.makedirs("/tmp/cli", exist_ok=True)
with open("/tmp/cli/q.fa", "wb") as f:
    f.write(b">query\n" + letters(q) + b"\n")
t0 = time.time()
all_letters = letters(flat)
with open("/tmp/cli/db.fa", "wb") as f:
    for i in range(n):
        s = all_letters[int(off[i]):int(off[i + 1])]
        f.write(b">sp|S%07d|SYNTH synthetic protein %d\n" % (i, i))
        f.write(b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n")
print("wrote %.1f MB FASTA in %.1f s" % (os.path.getsize("/tmp/cli/db.fa") / 1e6, time.time() - t0), flush=True)
cli = os.path.join(ROOT, "seq-align-gpu_amd", "bin", "smith_waterman")
mat = os.path.join(ROOT, "seq-align-gpu_amd", "data", "PAM250.txt")
for extra in ([], ["--topk", "100", "--align"], ["--savedb", "/tmp/cli/db.swg"], ["--packed"]):
    dbf = "/tmp/cli/db.swg" if "--packed" in extra else "/tmp/cli/db.fa"
    walls = []
    for rep in range(7):
        t0 = time.time()
        r = subprocess.run([cli, "--substitution_matrix", mat, "--timing"] + extra + ["--files", "/tmp/cli/q.fa", dbf],
                           stdout=open("/tmp/cli/out.txt", "wb"), stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, SWG_TIMING="1") if not extra else None)   # (plain run: swg_create's own breakdown too)
        walls.append((time.time() - t0) * 1e3)
    wall = sorted(walls[1:])[len(walls[1:]) // 2] / 1e3     # median of six runs after the first
    print("   walls of runs 2-7, ms: " + " ".join("%.0f" % w for w in walls[1:]))
    print("== %s: exit %d, wall %.0f ms (median of runs 2-7), stdout %.1f MB" % (" ".join(extra) or "plain", r.returncode, wall * 1e3,
                                                                    os.path.getsize("/tmp/cli/out.txt") / 1e6))
    print(r.stderr, flush=True)
    print(subprocess.run(["tail", "-2", "/tmp/cli/out.txt"], stdout=subprocess.PIPE, text=True).stdout if not extra else "", flush=True)
# what the dynamic loader does before main(): relocation statistics of the same command
r = subprocess.run([cli, "--help"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, LD_DEBUG="statistics"))
print("== LD_DEBUG=statistics (smith_waterman --help)")
print("\n".join(l for l in r.stderr.splitlines() if "total startup time" in l or "time needed for relocation" in l or "time needed to load objects" in l or "number of relocations:" in l))
t0 = time.time(); subprocess.run([cli, "--help"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print("wall of --help (load + static initialisers + exit): %.0f ms" % ((time.time() - t0) * 1e3))
print("host threads used by the library: %d; os.cpu_count %d; affinity %d" %
      (swg.lib.swg_host_threads(), os.cpu_count(), len(os.sched_getaffinity(0))))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
