#!/usr/bin/env python3
"""Tree-stage wall time: famsa-gpu vs the reference's own generators (oracle/_ref) on the same box.
Writes a JSON summary gpurun_out/e2e_<tag>.json (tag = argv[1], copied to profiles/).  Dev/measurement tool: the oracle is the baseline.
E2E_REFERENCE_FROM=<earlier summary>: do not run the reference again -- its times are copied from that file (same box type,
same script), and famsa-gpu's Newick is checked against the committed pins of the reference's output instead (tests/golden)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from famsa_amd import seqio

cli = os.path.join(ROOT, "famsa_amd", "famsa-gpu")
import multiprocessing as mp
threads = min(32, len(os.sched_getaffinity(0)))  # the reference's published setup; its spin barriers degrade beyond
TAG = sys.argv[1] if len(sys.argv) > 1 else "rXX"
OUT = os.path.join(ROOT, "gpurun_out", f"e2e_{TAG}.json")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
out = {"host_threads": threads, "cases": []}
PRIOR = {}
if os.environ.get("E2E_REFERENCE_FROM"):
    for c in json.load(open(os.environ["E2E_REFERENCE_FROM"]))["cases"]:
        PRIOR[(c["case"], c.get("gt") or c.get("mode"))] = c
    out["reference_times_from"] = os.environ["E2E_REFERENCE_FROM"]
META = json.load(open(os.path.join(ROOT, "tests", "golden", "meta_large.json")))


def _ref_worker(fasta, gt, heuristic, q):
    ref = oracle_bind.Ref()
    h = ref.open_fasta(fasta)
    t0 = time.time()
    want = ref.tree(h, gt, heuristic=heuristic, threads=threads)
    q.put((time.time() - t0, want))


def ref_tree(fasta, gt, heuristic, limit=150):
    q = mp.Queue()
    p = mp.Process(target=_ref_worker, args=(fasta, gt, heuristic, q))
    p.start()
    try:
        res = q.get(timeout=limit)
    except Exception:
        res = (None, None)
    p.join(timeout=1)
    if p.is_alive():
        p.kill()
    return res


WORK = []


REST_S = 6


def gpu(args, fasta, runs=2):
    """famsa-gpu `runs` times: (walls, tree-stage times, the Newick); WORK[-1] = the sums of the command's own stage timers
    (load, sort, upload, tree, Newick, store): wall - work = what starting a process that uses HIP costs (device discovery,
    context, the code objects' first use, exit), reported beside the rest so that a small case compares like with like"""
    walls, stages, text = [], [], None
    work = []
    for _ in range(runs):
        # the driver hands the device memory of a process that has ended back only after a while, and a process started
        # meanwhile waits for it (scripts/back_to_back.sh: -gt upgma at 100 000 sequences 1.6 s after a rest, 2.5-4 s right
        # after another such run): every run starts from a rested device
        # (6 s are enough after a small run; after one that held tens of GB the next 44 GB allocation takes 1-2 s for another
        #  ~20 s -- profiles/upgma_modes_r06.txt --, so the cases that allocate that much rest REST_S = 25 s)
        time.sleep(REST_S)
        t0 = time.time()
        p = subprocess.run([cli, "-v", *args, "-gt_export", fasta, "/tmp/cmp_gpu.dnd"], stderr=subprocess.PIPE, text=True)
        assert p.returncode == 0, p.stderr
        walls.append(time.time() - t0)
        kv = dict(l.split("=") for l in p.stderr.split() if "=" in l)
        stages.append(float(kv["time.tree_build"]))
        work.append(sum(float(kv.get("time." + k, 0)) for k in ("load", "sort", "gpu_upload", "tree_build", "newick", "store")))
        text = open("/tmp/cmp_gpu.dnd", "rb").read()
    WORK.append(work)
    return walls, stages, text


def case(name, fasta, gt, heuristic=0, cli_args=(), with_reference=True, limit=150, pin=None):
    """pin = sha256 of the reference's Newick for this input (tests/golden), or the golden file itself (bytes)"""
    import hashlib
    prior = PRIOR.get((name, gt))
    if prior is not None:
        t_ref, want = (prior["reference_tree_s"] if isinstance(prior["reference_tree_s"], float) else None), None
    else:
        t_ref, want = ref_tree(fasta, gt, heuristic, limit) if with_reference else (None, None)
    walls, stages, got = gpu(["-gt", gt, *cli_args], fasta)
    identical = (got == want) if want is not None else None
    if identical is None and pin is not None:
        identical = (got == pin) if isinstance(pin, bytes) else (hashlib.sha256(got).hexdigest() == pin)
    tb = min(stages)
    rec = {"case": name, "gt": gt, "identical_newick": identical,
           "identical_to": "the reference run beside it" if want is not None else ("the committed pin of the reference's output" if pin is not None else None),
           "reference_tree_s": round(t_ref, 3) if t_ref else ("> %d (stopped)" % limit if with_reference else "not run"),
           "gpu_tree_build_s": round(tb, 3), "gpu_cli_wall_s": round(min(walls), 3),
           "gpu_runs": [{"tree_build_s": round(a, 3), "cli_wall_s": round(b, 3), "stages_s": round(w, 3), "startup_s": round(b - w, 3)}
                        for a, b, w in zip(stages, walls, WORK[-1])],
           "gpu_startup_s": round(min(b - w for b, w in zip(walls, WORK[-1])), 3),
           "speedup_tree_stage": round(t_ref / tb, 1) if t_ref else None}
    print(rec, flush=True)
    out["cases"].append(rec)
    json.dump(out, open(OUT, "w"), indent=1)


codes, offsets = seqio.synth_uniform(10000, 400)
seqio.to_fasta(codes, offsets, "/tmp/cmp_10k.fasta")
HEMO = os.path.join(ROOT, "tests", "golden", "hemopexin")
case("synthetic 10000 x 400 aa", "/tmp/cmp_10k.fasta", "sl", pin=META["synth10k"]["sl_newick_sha256"])
case("synthetic 10000 x 400 aa", "/tmp/cmp_10k.fasta", "upgma", pin=META["synth10k"]["upgma_newick_sha256"])
case("synthetic 10000 x 400 aa", "/tmp/cmp_10k.fasta", "slink", pin=META["synth10k"]["slink_newick_sha256"])
case("hemopexin (4188 seqs, 21-210 aa)", os.path.join(HEMO, "hemopexin"), "nj", pin=open(os.path.join(HEMO, "nj.dnd"), "rb").read())
case("hemopexin (4188 seqs, 21-210 aa)", os.path.join(HEMO, "hemopexin"), "upgma", pin=open(os.path.join(HEMO, "upgma.dnd"), "rb").read())


def _ref_dist_worker(fasta, path, q):
    ref = oracle_bind.Ref()
    h = ref.open_fasta(fasta)
    t0 = time.time()
    ref.dist_export(h, path, threads=threads)
    q.put(time.time() - t0)


def dist_case(name, fasta):
    """-dist_export (lower triangle CSV): reference DistanceCalculator vs famsa-gpu, byte comparison."""
    import hashlib
    q = mp.Queue()
    p = mp.Process(target=_ref_dist_worker, args=(fasta, "/tmp/cmp_ref.csv", q))
    p.start()
    t_ref = q.get(timeout=300)
    p.join()
    t0 = time.time()
    r = subprocess.run([cli, "-v", "-dist_export", fasta, "/tmp/cmp_gpu.csv"], stderr=subprocess.PIPE, text=True)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()
    rec = {"case": name, "mode": "-dist_export", "identical_csv": md5("/tmp/cmp_ref.csv") == md5("/tmp/cmp_gpu.csv"),
           "csv_bytes": os.path.getsize("/tmp/cmp_gpu.csv"), "reference_s": round(t_ref, 3), "gpu_cli_wall_s": round(wall, 3),
           "speedup": round(t_ref / wall, 1)}
    print(rec, flush=True)
    out["cases"].append(rec)
    os.remove("/tmp/cmp_ref.csv")
    os.remove("/tmp/cmp_gpu.csv")


dist_case("hemopexin (4188 seqs)", os.path.join(ROOT, "tests", "golden", "hemopexin", "hemopexin"))
dist_case("synthetic 10000 x 400 aa", "/tmp/cmp_10k.fasta")
codes, offsets = seqio.synth_uniform(100000, 400)
seqio.to_fasta(codes, offsets, "/tmp/cmp_100k.fasta")
REST_S = 25
out["rest_before_each_large_run_s"] = REST_S
for gt in ("sl", "slink", "upgma"):
    case("synthetic 100000 x 400 aa", "/tmp/cmp_100k.fasta", gt, with_reference=False, pin=META["synth100k"].get(gt + "_newick_sha256"))
REST_S = 6
for n, ref_limit in ((200000, 150), (1000000, 240), (3000000, 0)):
    fam = "/tmp/family_%d_300.fasta" % n
    if not os.path.exists(fam):
        seqio.family_fasta(n, 300, fam)
    if os.path.exists(fam):
        case("synthetic family %d x ~255 aa, -medoidtree" % n, fam, "upgma", heuristic=2, cli_args=["-medoidtree"],
             with_reference=ref_limit > 0, limit=ref_limit, pin=META["family%d" % n]["medoid_upgma_newick_sha256"])
json.dump(out, open(OUT, "w"), indent=1)
