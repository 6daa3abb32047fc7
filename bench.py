#!/usr/bin/env python3
"""bench.py -- all-pairs bit-parallel LCS throughput on MI355X (the metric of BASELINE.json).

One "step" = one full pass of the hot path over the synthetic set already resident in HBM:
every pair (ref = row i, partner = column j < i) of the lower triangle -> integer LCS (uint16,
left in HBM), then the single-linkage reduction over that triangle down to the minimum spanning
tree MSTPrim would build (n-1 edges on the host, in Prim's insertion order).
With N > 1 ranks (one process per GPU, torch.distributed over RCCL) the rows are split into N
row blocks of equal pair count -- no data-path collective for the LCS itself -- and the tree is
found by Boruvka rounds over the row-block-resident triangles: per round one all-gather of
n x 16 bytes per rank (every vertex's best edge into another component among the rank's pairs;
round 0 = the per-row minima completed by the per-column minima), component labels replicated.
The total work is fixed, so the scaling is "strong".  The edge-list hash printed in `mst` is the
same for every N.

    python bench.py [--gpus N --steps K --warmup W] [--n-seqs 100000 --seq-len 400]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload family|realmix [--order sorted|input]     ragged lengths (below)
    python bench.py --pmc off                                            without the two profiled passes behind roofline.traffic (below)

Workloads.  The default (`uniform`: BASELINE.json's configs[3], 100 000 x 400 aa) and its line are what they always
were.  `--workload family` (one ancestor of --seq-len 300 residues, 25 % substitutions, lengths uniform in [0.7 L, L]:
the C5 shape) and `--workload realmix` (the upstream real sets held under tests/golden as one 13 774-record set,
21-210 residues) measure the same step on ragged inputs, `--order sorted` in FAMSA's working order (length
descending: what every tree generator uploads) or `--order input` as read (what -dist_export uploads).  Their line
has the same fields; cells, algorithmic bytes and VALU operations are summed over the actual lengths
(sum over pairs of len_partner x ceil(len_ref / 32) x 3 lane-ops), `config.workload` names the set and the order.

`roofline.traffic` (one rank): after the timed loop this command is run again twice under `rocprofv3 --kernel-trace --pmc
FETCH_SIZE` resp. `WRITE_SIZE` (one step each, separate passes as MI355X_MICROARCH.md prescribes; ~15 s each), and
`roofline.traffic` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the LCS kernels of the step (gfx950: FETCH_SIZE counts
half of a wide streaming read).  By default whenever rocprofv3 is on PATH (`--pmc auto`); with `--pmc off`, with more than one
rank, or without rocprofv3 `traffic` is null -- it is never imported from a file.

Both forms work for N > 1: started plainly (no WORLD_SIZE in the environment), `bench.py --gpus N` starts its
N ranks itself (one process per GPU, LOCAL_RANK binding, rendezvous on 127.0.0.1) and relays rank 0's line.
With more than one rank a fail-closed SELF-CHECK runs before the timed loop: a 2000-sequence family set, the
sharded tree of all ranks against the single-context tree every rank computes alone -- any difference, HIP
or RCCL error ends the run with the rank, the transport and the error text instead of a number.
`--mode contexts`: ONE process, N engine contexts (one per GPU) driven through lcsgpu_multi_mst_prim -- the
library's own multi-GPU path behind `famsa-gpu -gpu a,b,..` (key exchange: grouped ncclAllGather when every
context has its own device, else peer copies).

Rank 0 prints ONE JSON line.  After the timed region a sample of the triangle the last step left
in HBM is compared with the oracle (oracle/lcs_oracle.c) -- `parity`.  `cpu_baseline` times the
REFERENCE's own AVX2 path (oracle/_ref/libfamsa_ref.so = /root/reference sources +
oracle/ref_harness.cpp) on a bounded sample of the same workload on this host's cores; the
oracle is only the checker / the reported baseline and is never on the measured GPU path.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PAIR_EXTRA = 2  # uint16 result; + len_partner residue bytes
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# integer-VALU issue peak: 256 CUs x 4 SIMD-32 x 32 lanes per clock at the nominal 2.4 GHz
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9


def cpu_quota():
    try:  # cgroup v2 CPU quota of the container, if any
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return int(q) / int(period)
    except Exception:
        pass
    return None


def cpu_baseline(n, length, target_s=12.0, codes=None, offsets=None):
    """The reference's UPGMA::computeDistances (tree/UPGMA.cpp:75-109, AVX2 dispatch) on the first
    n_use sequences of the same synthetic set."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    from famsa_amd import seqio
    if not oracle_bind.have_ref():
        return None
    ref = oracle_bind.Ref()
    avail = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    n_probe = min(n, 4000)
    ragged = codes is not None
    if not ragged:
        codes, offsets = seqio.synth_uniform(n, length)
    # the sample is a prefix of the same set (fixed length => already in the reference's order up to ties)
    n_load = min(n, 60000)
    path = f"/tmp/bench_synth_{n}_{length}_{n_load}.fasta"
    seqio.to_fasta(codes[: int(offsets[n_load])], offsets[: n_load + 1], path)
    h = ref.open_fasta(path)
    # The reference's row queue does not scale to every hardware thread of a big host, and a container
    # CPU quota lets short runs burst far above what it sustains: time ~target_s/3 of work per candidate
    # thread count (the quota, twice the quota, and the best of a quick probe) and report the fastest.
    cand = {16, 32, 64, 128, avail}
    if quota:
        cand.add(max(1, int(round(quota))))
    probe = {}
    for threads in sorted(t for t in cand if t <= avail):
        sec, pairs, cells, _ = ref.time_triangle(h, n_probe, threads)
        probe[threads] = pairs / sec
    finalists = {max(probe, key=probe.get)}
    if quota:
        finalists |= {t for t in (max(1, int(round(quota))), 2 * int(round(quota))) if t <= avail}
    sustained = {}
    best = None
    for threads in sorted(finalists):
        n_use = int(min(n_load, max(n_probe, math.sqrt(2 * probe.get(threads, max(probe.values())) * target_s / 3))))
        sec, pairs, cells, _ = ref.time_triangle(h, n_use, threads)
        sustained[threads] = pairs / sec
        if best is None or pairs / sec > best[1] / best[0]:
            best = (sec, pairs, cells, threads, n_use)
    sec, pairs, cells, threads, n_use = best
    ref.close(h)
    os.unlink(path)
    return {
        "value": cells / sec / 1e9,
        "unit": "Gcell/s",
        "cores": threads,
        "kind": "reference",
        "pairs_per_s": pairs / sec,
        "seconds": sec,
        "host_threads_available": avail,
        "host_cpu_quota_cores": quota,
        "probe_pairs_per_s_by_threads": {str(t): r for t, r in sorted(probe.items())},
        "sustained_pairs_per_s_by_threads": {str(t): r for t, r in sorted(sustained.items())},
        "sample": ("(the reference sorts its input by length first: the sample is the LONGEST sequences of the set) " if ragged else "") +
                  f"reference UPGMA::computeDistances (AVX2 dispatch, {threads} threads = the fastest sustained of "
                  f"{'/'.join(str(t) for t in sorted(sustained))}; the container's CPU quota is "
                  f"{quota if quota else 'unlimited'} cores of {avail} hardware threads) on the first {n_use} of "
                  f"the {n} synthetic sequences = {int(pairs)} pairs",
    }


def make_workload(args):
    """(codes, offsets, description, uniform) of the set the steps run on."""
    from famsa_amd import seqio
    n, L = args.n, args.len
    if args.workload == "uniform":
        codes, offsets = seqio.synth_uniform(n, L)
        return codes, offsets, None, True
    if args.workload == "family":
        codes, offsets = seqio.family_set(n, L)
        what = f"synthetic family of {n} proteins (one ancestor of {L} aa, 25 % substitutions, lengths uniform in [{int(L * 0.7)}, {L}])"
    else:
        import famsa_amd
        golden = os.path.join(ROOT, "tests", "golden")
        seqs = []
        for _, rel in seqio.REALMIX_PARTS:
            seqs += [famsa_amd.lcsgpu.encode(r) for r in seqio.read_fasta(os.path.join(golden, rel))[1]]
        codes, offsets = seqio.pack(seqs)
        what = f"the upstream real sets as one input ({len(seqs)} records, {min(map(len, seqs))}-{max(map(len, seqs))} aa, duplicates kept)"
    if args.order == "sorted":
        o = offsets.astype(np.int64)
        order = seqio.sort_order([codes[o[i]:o[i + 1]] for i in range(len(o) - 1)])
        codes, offsets = seqio.reorder(codes, offsets, np.asarray(order, dtype=np.int64))
        what += ", in FAMSA's working order (length descending, then residues)"
    else:
        what += ", in input order"
    return codes, offsets, what, False


def block_sums(offsets, r0, r1):
    """Over the pairs (i, j < i) of the rows [r0, r1): cells = sum len_i len_j, algorithmic bytes = sum (len_j + 2),
    VALU lane-ops = sum 3 len_j ceil(len_i / 32) -- exact integers as floats."""
    lens = np.diff(offsets.astype(np.int64)).astype(np.float64)
    before = np.concatenate([[0.0], np.cumsum(lens)[:-1]])  # residues of the partners j < i
    rows = np.arange(len(lens), dtype=np.float64)
    sl = slice(r0, r1)
    H = np.ceil(lens[sl] / 32.0)
    return {"cells": float((lens[sl] * before[sl]).sum()), "bytes": float((before[sl] + 2.0 * rows[sl]).sum()),
            "valu_ops": float((3.0 * H * before[sl]).sum()), "pairs": float(rows[sl].sum()),
            "classes": sorted({int(h) for h in H[lens[sl] > 0]})}


def measure_traffic(argv):
    """roofline.traffic measured by THIS run (--pmc): the command again, one step, under rocprofv3 with FETCH_SIZE resp.
    WRITE_SIZE (separate passes, --kernel-trace only: MI355X_MICROARCH.md, HBM / rocprofv3 section); per step the sum over
    the LCS kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950: FETCH_SIZE counts half of a wide streaming read).
    Returns (bytes or None, how / why not)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    keep, skip = [], False
    for a in argv:  # the command as it was given, without its own --pmc [value]
        if skip and a in ("auto", "on", "off"):
            skip = False
            continue
        skip = a == "--pmc"
        if not skip and not a.startswith("--pmc="):
            keep.append(a)
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "run", "--", sys.executable, os.path.abspath(__file__)] + keep + \
              ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-parity", "--pmc", "off"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=300, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            total = 0.0
            for path in dbs:
                cur = sqlite3.connect(path).cursor()
                for name, value in cur.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? "
                                               "group by kernel_name", (counter,)):
                    if "lcs_rows_kernel" in name or "lcs_long_kernel" in name:
                        total += float(value)
            if not dbs or total <= 0:
                return None, f"the rocprofv3 pass for {counter} recorded nothing for the LCS kernels"
            sums[counter] = total
        except Exception as e:  # a report, never a reason to lose the line
            return None, f"the rocprofv3 pass for {counter} failed: {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0, \
        "measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one step each, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over the step's LCS kernels"


def transport_report(torch, world, rank, local_rank, emulate):
    """What the ranks will reach each other over, before anything is timed: peer access between all device pairs, the
    RCCL version torch was built with, the NCCL / RCCL / HSA settings in the environment, the xGMI topology as rocm-smi
    prints it (rank 0, bounded)."""
    rep = {"rank": rank, "local_rank": local_rank, "devices_visible": torch.cuda.device_count()}
    try:
        nd = torch.cuda.device_count()
        rep["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(nd)] for i in range(nd)]
    except Exception as e:
        rep["peer_access"] = f"unavailable: {e}"
    try:
        rep["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:
        rep["rccl_version"] = f"unavailable: {e}"
    rep["env"] = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "LCSGPU_"))}
    if rank == 0 and not emulate:
        try:
            import subprocess
            p = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=20)
            rep["rocm_smi_topology"] = [ln for ln in p.stdout.splitlines() if ln.strip() and not ln.startswith("=")][:24]
        except Exception as e:
            rep["rocm_smi_topology"] = f"unavailable: {type(e).__name__}"
    return rep


class Watchdog:
    """A phase that does not end within `seconds` ends the process with a message that names the rank and the phase
    (a hung rendezvous or collective otherwise looks like a silent stall of the whole job)."""
    def __init__(self, seconds, what):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.seconds = what, seconds
    def _fire(self):
        print(f"bench.py: {self.what} did not finish within {self.seconds:.0f} s -- giving up", file=sys.stderr, flush=True)
        os._exit(5)
    def __enter__(self):
        self.t.start()
        return self
    def __exit__(self, *exc):
        self.t.cancel()
        return False


def sampled_parity(eng, tri, r0, r1, codes, offsets, n_samples=6000, seed=11):
    """Compare a sample of the rank's triangle in HBM (what the timed steps produced) with the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    import torch
    if r1 <= max(r0, 1):
        return 0, 0
    oracle = oracle_bind.Oracle()
    rng = np.random.Generator(np.random.PCG64(seed + r0))
    rows = rng.integers(max(r0, 1), r1, size=n_samples)
    rows[:4] = [max(r0, 1), r1 - 1, max(r0, 1), r1 - 1]
    cols = (rng.random(n_samples) * rows).astype(np.int64)
    cols[:2] = [0, 0]
    cols[2:4] = rows[2:4] - 1
    off = r0 * (r0 - 1) // 2
    idx = rows.astype(np.int64) * (rows - 1) // 2 + cols - off
    got = tri[torch.from_numpy(idx).to(tri.device)].cpu().numpy().astype(np.int64) & 0xFFFF
    o = offsets.astype(np.int64)
    bad = 0
    for k in range(n_samples):
        i, j = int(rows[k]), int(cols[k])
        want = oracle.lcs(codes[o[i]:o[i + 1]], codes[o[j]:o[j + 1]])  # ref = row, partner = column
        bad += int(want != int(got[k]))
    return n_samples, bad


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly: start the N ranks (what torch.distributed.run would do -- one
    process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), wait for them, relay rank 0's
    JSON line.  The first rank that fails ends the others (by their PIDs) and the exit code says so."""
    import subprocess
    import famsa_amd
    n = args.gpus
    have = famsa_amd.load_library().lcsgpu_device_count()
    if not args.emulate_ranks_on_one_gpu and have < n:
        raise SystemExit(f"bench.py --gpus {n}: {have} GPU(s) visible on this host, {n} needed (one rank per GPU).  "
                         f"--emulate-ranks-on-one-gpu runs the {n}-rank path functionally on one GPU (gloo exchange).")
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    failed = None
    alive = set(range(n))
    while alive and failed is None:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    if failed is not None:
        for r in alive:
            procs[r].terminate()
        for r in alive:
            try:
                procs[r].wait(timeout=20)
            except Exception:
                procs[r].kill()
        print(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}; the other ranks were stopped", file=sys.stderr)
        raise SystemExit(failed[1] if failed[1] > 0 else 1)


def self_check(eng, torch, dist, rank, world, dev, make_exchange, transport):
    """Fail closed before any number is produced with more than one rank: on a 2000-sequence family set (ragged
    lengths, near ties) every rank builds the tree alone (lcsgpu_mst_prim, the single-context path the oracle's
    recurrence pins in the test suite) and the ranks build it together over the same exchange the timed loop uses;
    every rank's sharded edge-list hash must equal every rank's single-context hash."""
    from famsa_amd import seqio
    from famsa_amd.rowblock import row_cuts, pairs_in_rows, sharded_mst_device, edge_list_sha256
    t0 = time.perf_counter()
    try:
        seqs = seqio.synth_family(2000, 160, seed=0xC4EC)
        eng.upload_seqs(seqs)
        n = len(seqs)
        alone = edge_list_sha256(eng.mst_prim(1))
        cuts = row_cuts(n, world)
        r0, r1 = cuts[rank], cuts[rank + 1]
        tri = torch.empty(max(pairs_in_rows(r0, r1), 1), dtype=torch.int16, device=dev)
        keys, gathered, all_gather = make_exchange(n)
        torch.cuda.synchronize()
        eng.lcs_triangle_dev(r0, r1, tri.data_ptr(), 2)
        edges, rounds = sharded_mst_device(eng, tri.data_ptr(), 2, r0, r1, 1, keys, gathered, all_gather)
        together = edge_list_sha256(edges)
        eng.sync()
        rec = [None] * world
        if world > 1 or dist is not None:
            dist.all_gather_object(rec, (alone, together))
        else:
            rec = [(alone, together)]
    except BaseException as e:  # HIP / RCCL / rendezvous errors included: say where and over what, then stop
        print(f"bench.py SELF-CHECK FAILED on rank {rank} of {world} (transport: {transport}): {type(e).__name__}: {e}",
              file=sys.stderr, flush=True)
        os._exit(3)
    hashes = {h for pair in rec for h in pair}
    if len(hashes) != 1:
        lines = "; ".join(f"rank {r}: alone {a[:12]} together {t[:12]}" for r, (a, t) in enumerate(rec))
        print(f"bench.py SELF-CHECK FAILED on rank {rank} of {world} (transport: {transport}): the sharded tree differs from the "
              f"single-context tree -- {lines}", file=sys.stderr, flush=True)
        os._exit(3)
    return {"n_seqs": n, "rounds": int(rounds), "edges_sha256": alone, "ranks_agree": True, "seconds": time.perf_counter() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-seqs", dest="n", type=int, default=100000)
    ap.add_argument("--seq-len", dest="len", type=int, default=400)
    ap.add_argument("--workload", choices=["uniform", "family", "realmix"], default="uniform",
                    help="uniform (default): --n-seqs x --seq-len, BASELINE.json's config; family: ragged lengths in [0.7 L, L] "
                         "(the C5 shape; --seq-len defaults to 300 then); realmix: the 13 774 upstream real sequences")
    ap.add_argument("--order", choices=["sorted", "input"], default="sorted",
                    help="ragged workloads: FAMSA's working order (length descending) or the order as read")
    ap.add_argument("--pmc", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic: this command again under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one step each, after the "
                         "timed loop (one rank).  auto (default): when rocprofv3 is on PATH; on: report why if it cannot; off: traffic = null")
    ap.add_argument("--phase-timeout-s", type=float, default=600.0,
                    help="rendezvous, self-check and every other untimed phase with more than one rank: give up after this long, naming the rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--mode", choices=["ranks", "contexts"], default="ranks",
                    help="ranks (default): one process per GPU, torch.distributed over RCCL, the lcsgpu_mst_shard_* protocol; "
                         "contexts: ONE process, one engine context per GPU, lcsgpu_multi_mst_prim (the library's own "
                         "multi-GPU single linkage: what famsa-gpu -gpu a,b,.. runs)")
    ap.add_argument("--emulate-ranks-on-one-gpu", action="store_true",
                    help="functional test of the N>1 path on a 1-GPU box: all ranks / contexts use cuda:0 (ranks: gloo exchange)")
    ap.add_argument("--mst-mode", choices=["fused", "passes"], default="passes",
                    help="passes (default): plain LCS launch, every Boruvka round streams the resident triangle; "
                         "fused: the LCS launch does round 0's local half on the values it holds in registers "
                         "(lcsgpu_mst_shard_begin, LCSGPU_MST_COMPUTE) -- measured slower by 16 ms per step at 100 000 x "
                         "400 aa (the fold's registers cost the launch more than the saved passes), kept for A/B runs")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: initialise the nccl (RCCL) backend anyway and run the real "
                         "all_gather_into_tensor + stream hand-off every Boruvka round, as the N>1 path does")
    ap.add_argument("--self-check", action="store_true",
                    help="run the multi-rank self-check also with one rank (it always runs with more than one)")
    args = ap.parse_args()
    if args.workload == "family" and "--seq-len" not in sys.argv:
        args.len = 300

    if args.mode == "contexts":
        if args.workload != "uniform":
            raise SystemExit("bench.py --mode contexts runs the uniform workload only")
        return main_contexts(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)

    import torch
    import famsa_amd
    from famsa_amd import seqio
    from famsa_amd.rowblock import row_cuts, pairs_in_rows, sharded_mst_device, edge_list_sha256
    from famsa_amd.lcsgpu import MST_COMPUTE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch with "
                         f"--nproc-per-node {args.gpus}, or plainly (python bench.py --gpus {args.gpus} starts its ranks itself)")
    dist = None
    emulate = args.emulate_ranks_on_one_gpu
    if emulate:
        local_rank = 0
    collective = world > 1 or args.force_collective  # the exchange runs as a real collective
    transport = "none (one block)"
    rccl_ranks = None
    if collective:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        rendezvous = Watchdog(args.phase_timeout_s, f"rank {rank} of {world}: the rendezvous on {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}")
        rendezvous.__enter__()
        try:
            if emulate:
                transport = f"gloo (host memory), {world} ranks sharing cuda:0"
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.phase_timeout_s))
            else:
                n_dev = torch.cuda.device_count()
                if local_rank >= n_dev:
                    raise RuntimeError(f"LOCAL_RANK {local_rank} but {n_dev} GPU(s) visible")
                transport = f"nccl = RCCL all_gather_into_tensor (device memory), {world} ranks, rank {rank} on cuda:{local_rank}"
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=args.phase_timeout_s))
                rccl_ranks = dist.get_world_size()
        except BaseException as e:
            print(f"bench.py: rank {rank} of {world} could not join the process group (transport: {transport}): "
                  f"{type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(4)
        rendezvous.__exit__()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    report = transport_report(torch, world, rank, local_rank, emulate) if world > 1 or args.force_collective or args.self_check else None
    if report is not None and rank == 0:  # before anything is timed: what the ranks will talk over
        print("bench.py transport report: " + json.dumps(report), file=sys.stderr, flush=True)

    n, L = args.n, args.len
    eng = famsa_amd.LcsGpu(local_rank)
    ext = torch.cuda.ExternalStream(eng._lib.lcsgpu_stream(eng._ctx), device=dev)

    def make_exchange(n_keys):
        """(keys, gathered, all_gather) for a set of n_keys sequences: this rank's lcsgpu_mst_key records, every rank's,
        and the call that fills the latter from the former in the right stream order."""
        keys = torch.zeros(2 * n_keys, dtype=torch.int64, device=dev)
        gathered = torch.zeros(world * 2 * n_keys, dtype=torch.int64, device=dev) if collective else keys
        if not collective:
            def all_gather(g, k):
                pass                                   # one block: its keys are the gathered keys
        elif not emulate:
            def all_gather(g, k):
                cur = torch.cuda.current_stream()
                cur.wait_stream(ext)                   # the keys are produced on the engine's stream
                dist.all_gather_into_tensor(g, k)      # RCCL over xGMI: n x 16 B per rank
                ext.wait_stream(cur)                   # the merge kernels read the gathered keys
        else:
            def all_gather(g, k):
                eng.sync()
                h = torch.empty(g.shape, dtype=g.dtype)
                dist.all_gather_into_tensor(h, k.cpu())
                g.copy_(h)
                torch.cuda.synchronize()
        return keys, gathered, all_gather

    check = None
    if world > 1 or args.self_check:
        with Watchdog(args.phase_timeout_s, f"rank {rank} of {world}: the self-check (transport: {transport})"):
            check = self_check(eng, torch, dist if collective else None, rank, world, dev, make_exchange, transport)

    codes, offsets, workload_text, uniform = make_workload(args)
    n = len(offsets) - 1
    eng.upload(codes, offsets)  # inputs resident in HBM before the timed region

    cuts = row_cuts(n, world)
    r0, r1 = cuts[rank], cuts[rank + 1]
    my_pairs = pairs_in_rows(r0, r1)
    total_pairs = n * (n - 1) // 2
    tri = torch.empty(max(my_pairs, 1), dtype=torch.int16, device=dev)
    keys, gathered, all_gather = make_exchange(n)
    torch.cuda.synchronize()  # torch zero-fills on ITS stream; the engine writes these buffers on its own

    kernel_ms, mst_ms = [], []
    last = {}
    from famsa_amd.lcsgpu import mst_order_edges

    def finish_previous():
        """Prim's insertion order of the previous step's tree (a walk over its n-1 edges on the host, ~10 ms at
        100 000 sequences): done while the GPU computes the next step's triangle, as a pipeline over consecutive
        problems would; the last step's is done before the clock stops."""
        if "unordered" in last:
            last["edges"] = mst_order_edges(last.pop("unordered"), n)

    fused = args.mst_mode == "fused"

    def step():
        if fused:  # one launch: LCS values -> uint16 triangle in HBM + every vertex's best edge (round 0's local half)
            eng.mst_shard_begin(tri.data_ptr(), 2, r0, r1, 1 | MST_COMPUTE)
        else:
            eng.lcs_triangle_dev(r0, r1, tri.data_ptr(), 2)
        finish_previous()
        ms, _ = eng.last_kernel_ms()  # HIP events on the engine's stream, around the LCS launch (waits for it)
        kernel_ms.append(ms)
        t0 = time.perf_counter()
        edges, rounds = sharded_mst_device(eng, tri.data_ptr(), 2, r0, r1, 1, keys, gathered, all_gather, ordered=False,
                                           begun=fused)
        mst_ms.append((time.perf_counter() - t0) * 1e3)
        last["unordered"], last["rounds"] = edges, rounds

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish_previous()
    fence()
    kernel_ms.clear()
    mst_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish_previous()
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if emulate else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: the result of the last timed step against the oracle (a sample of this
    # rank's triangle), and every rank must have ended in the same tree
    sampled, bad = (0, 0) if args.no_parity else sampled_parity(eng, tri, r0, r1, codes, offsets)
    edges = last["edges"]
    edges_hash = edge_list_sha256(edges)
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    per_rank_kernel_ms = [k_ms]
    if collective:
        rec = [None] * world
        dist.all_gather_object(rec, (edges_hash, sampled, bad, k_ms))
        assert len({h for h, _, _, _ in rec}) == 1, "ranks disagree on the tree"
        sampled, bad = sum(s for _, s, _, _ in rec), sum(b for _, _, b, _ in rec)
        per_rank_kernel_ms = [k for _, _, _, k in rec]
    assert bad == 0, f"{bad} of {sampled} sampled pairs differ from the oracle"
    assert len(edges) == n - 1 and (edges["from"] < edges["to"]).all()

    if rank == 0:
        exchange = ("gloo (host memory)" if emulate else "nccl = RCCL (device memory)") if collective else "none (one block)"
        sums = None if uniform else {"mine": block_sums(offsets, r0, r1), "all": block_sums(offsets, 0, n), "text": workload_text,
                                         "mean_len": float(np.diff(offsets.astype(np.int64)).mean())}
        out = result_line(args, n, L, world, elapsed, k_ms, my_pairs, total_pairs, fused, sums,
                          parallelism=f"rowblock{world}" + ("+allgather(n x 16 B best-edge keys per Boruvka round)" if collective else ""),
                          exchange=exchange, edges=edges, edges_hash=edges_hash, rounds=last["rounds"], mst_ms=float(np.mean(mst_ms)),
                          sampled=sampled, bad=bad, library=eng._lib.lcsgpu_version().decode())
        out["ranks"] = {"mode": "ranks", "world": world, "rccl_ranks": rccl_ranks, "transport": transport, "transport_report": report,
                        "kernel_ms_per_rank": per_rank_kernel_ms, "kernel_ms_min": float(min(per_rank_kernel_ms)),
                        "kernel_ms_max": float(max(per_rank_kernel_ms)), "self_check": check}
        import shutil
        if world == 1 and (args.pmc == "on" or (args.pmc == "auto" and shutil.which("rocprofv3"))):
            eng.close()  # the profiled child gets the device to itself
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measure_traffic(sys.argv[1:])
        if not args.no_cpu_baseline and world == 1:
            try:
                cb = cpu_baseline(n, L, codes=None if uniform else codes, offsets=None if uniform else offsets)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                cb = {"error": repr(e)}
            out["cpu_baseline"] = cb
        emit(out)
    if collective:
        dist.destroy_process_group()


def emit(out):
    try:  # RCCL writes its version banner through C stdio: out with it before the one JSON line, not after
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def result_line(args, n, L, world, elapsed, k_ms, my_pairs, total_pairs, fused, sums, parallelism, exchange, edges, edges_hash, rounds,
                mst_ms, sampled, bad, library):
    ms_per_step = elapsed / args.steps * 1e3
    if sums is None:  # uniform lengths: closed forms
        cells = float(total_pairs) * L * L
        H = (L + 31) // 32
        algo_bytes = my_pairs * (L + ALGO_BYTES_PER_PAIR_EXTRA)
        ops_per_pair = 3 * L * H  # three VALU lane-ops per partner residue and 32-bit half-word of the ref
        my_ops = my_pairs * ops_per_pair
        bytes_per_pair = L + ALGO_BYTES_PER_PAIR_EXTRA
        kernel = f"lcsgpu::lcs_rows_kernel_pipe<{H}, {4 if H <= 16 else 2 if H <= 32 else 1}, 4, {'true' if fused else 'false'}>"
        workload = (f"synthetic {n} proteins x {L} aa (uniform over 20 residues), full all-pairs LCS "
                    f"lower triangle -> uint16 in HBM -> single-linkage MST (n-1 edges, Prim's order) on the host")
        config_extra = {"n_seqs": n, "seq_len": L, "pairs": total_pairs}
    else:  # ragged lengths: sums over the actual pairs
        cells = sums["all"]["cells"]
        algo_bytes = sums["mine"]["bytes"]
        my_ops = sums["mine"]["valu_ops"]
        ops_per_pair = my_ops / max(sums["mine"]["pairs"], 1.0)
        bytes_per_pair = algo_bytes / max(sums["mine"]["pairs"], 1.0)
        kernel = ("lcsgpu::lcs_rows_kernel_pipe<H, ..> for the half-word classes H = " + ",".join(str(h) for h in sums["mine"]["classes"]) +
                  " of the refs (small neighbouring classes share a launch), one after the other")
        workload = (f"{sums['text']}, full all-pairs LCS lower triangle -> uint16 in HBM -> single-linkage MST (n-1 edges, "
                    f"Prim's order) on the host")
        config_extra = {"n_seqs": n, "seq_len": L, "pairs": total_pairs, "order": args.order, "mean_len": sums["mean_len"]}
    value = cells * args.steps / elapsed / 1e9
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    valu_rate = my_ops / (k_ms * 1e-3)
    traffic, traffic_source = None, "not measured (--pmc off, more than one rank, or no rocprofv3 on PATH)"
    return {
        "metric": "lcs_gcell_updates_per_s",
        "value": value,
        "unit": "Gcell/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            **config_extra,
            "parallelism": parallelism,
            "exchange": exchange,
        },
        "library": library,
        "pairs_per_s": total_pairs * args.steps / elapsed,
        "mst": {"mode": args.mst_mode, "n_edges": int(len(edges)), "rounds": None if rounds is None else int(rounds), "edges_sha256": edges_hash,
                "ms_per_step": mst_ms, "exchange_bytes_per_rank_per_round": 16 * n},
        "parity": {"sampled_pairs": int(sampled), "mismatches": int(bad),
                   "checker": "oracle/lcs_oracle.c on a sample of the triangle the last timed step left in HBM"},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # HBM-side bytes per step's LCS launches: measured by this run with --pmc (measure_traffic), else null
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": kernel,
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_pair": bytes_per_pair,
            "pairs_per_launch": my_pairs,
            "note": "the contract figure: algorithmic bytes / kernel time against the HBM peak; the kernel is "
                    "integer-VALU bound by construction (SURVEY 8d), `valu` is the roof that binds",
            "valu": {
                "ops_per_pair": ops_per_pair,
                "achieved_ops_per_s": valu_rate,
                "peak_ops_per_s": VALU_PEAK_LANE_OPS,
                "frac": valu_rate / VALU_PEAK_LANE_OPS,
                "unit": "32-bit lane-ops/s; peak = 256 CU x 4 SIMD-32 x 32 lanes x 2.4 GHz nominal",
            },
        },
    }


def main_contexts(args):
    """--mode contexts: one process, N engine contexts (context k on GPU k), one lcsgpu_multi_mst_prim per step: every
    context computes its row block of the LCS triangle into its own HBM and the local half of each Boruvka round, the
    keys are exchanged device to device (the library picks: one grouped ncclAllGather per round when every context has
    its own device, else peer copies), every context runs the same global half, context 0's edge list comes back in
    Prim's order.  The product path behind `famsa-gpu -gpu 0,1,..`; no torch.distributed."""
    import torch
    import famsa_amd
    from famsa_amd import seqio
    from famsa_amd.lcsgpu import LcsGpuGroup
    from famsa_amd.rowblock import edge_list_sha256
    N = args.gpus
    have = famsa_amd.load_library().lcsgpu_device_count()
    if args.emulate_ranks_on_one_gpu:
        devices = [0] * N
    elif have < N:
        raise SystemExit(f"bench.py --mode contexts --gpus {N}: {have} GPU(s) visible, {N} needed "
                         f"(--emulate-ranks-on-one-gpu puts all contexts on cuda:0)")
    else:
        devices = list(range(N))
    group = LcsGpuGroup(devices)
    n, L = args.n, args.len
    try:
        if N > 1 or args.self_check:  # fail closed, as in the ranks mode: the contexts' tree against one context's
            t0 = time.perf_counter()
            seqs = seqio.synth_family(2000, 160, seed=0xC4EC)
            group.upload_seqs(seqs)
            alone = edge_list_sha256(group.engs[0].mst_prim(1))
            together = edge_list_sha256(group.mst_prim(1))
            if alone != together:
                raise RuntimeError(f"the {N}-context tree ({together[:12]}) differs from the single-context tree ({alone[:12]})")
            check = {"n_seqs": len(seqs), "edges_sha256": alone, "contexts_agree": True, "seconds": time.perf_counter() - t0}
        else:
            check = None
    except BaseException as e:
        print(f"bench.py SELF-CHECK FAILED ({N} contexts on devices {devices}; {group.transport()!r}): {type(e).__name__}: {e}",
              file=sys.stderr, flush=True)
        os._exit(3)
    codes, offsets = seqio.synth_uniform(n, L)
    group.upload(codes, offsets)
    total_pairs = n * (n - 1) // 2
    kernel_ms, edges = [], None
    for _ in range(args.warmup):
        group.mst_prim(1)
    for e in group.engs:
        e.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        edges = group.mst_prim(1)
        kernel_ms.append(group.last_kernel_ms())
    for e in group.engs:
        e.sync()
    elapsed = time.perf_counter() - t0
    per_ctx = [float(np.mean([k[c] for k in kernel_ms])) for c in range(N)]
    # parity: the whole triangle is inside the library in this mode; check a sample of pairs through the same contexts' LCS path
    sampled = bad = 0
    if not args.no_parity:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_bind
        oracle = oracle_bind.Oracle()
        rng = np.random.Generator(np.random.PCG64(11))
        rows = rng.integers(1, n, size=64).astype(np.int32)
        cols = np.sort(rng.integers(0, n, size=96)).astype(np.int32)
        o = offsets.astype(np.int64)
        for c, eng in enumerate(group.engs):
            got = eng.lcs_rect(rows, cols)
            for a in range(0, len(rows), 7):
                for b in range(c, len(cols), 5):
                    i, j = int(rows[a]), int(cols[b])
                    sampled += 1
                    bad += int(oracle.lcs(codes[o[i]:o[i + 1]], codes[o[j]:o[j + 1]]) != int(got[a, b]))
    assert bad == 0, f"{bad} of {sampled} sampled pairs differ from the oracle"
    assert len(edges) == n - 1 and (edges["from"] < edges["to"]).all()
    my_pairs = total_pairs // N
    out = result_line(args, n, L, N, elapsed, float(max(per_ctx)), my_pairs, total_pairs, False, None,
                      parallelism=f"rowblock{N}, one process, {N} contexts (lcsgpu_multi_mst_prim)",
                      exchange=group.transport().splitlines()[-1], edges=edges, edges_hash=edge_list_sha256(edges), rounds=None,
                      mst_ms=float("nan"), sampled=sampled, bad=bad, library=group._lib.lcsgpu_version().decode())
    out["mst"]["ms_per_step"] = elapsed / args.steps * 1e3 - float(max(per_ctx))  # everything of a step that is not the slowest block's LCS launch
    out["parity"]["checker"] = "oracle/lcs_oracle.c on sampled pairs recomputed through every context (the triangles stay inside the library in this mode)"
    out["ranks"] = {"mode": "contexts", "world": N, "devices": devices, "transport": group.transport(), "rccl_ranks": None,
                    "kernel_ms_per_rank": per_ctx, "kernel_ms_min": float(min(per_ctx)), "kernel_ms_max": float(max(per_ctx)),
                    "self_check": check}
    emit(out)
    group.close()


if __name__ == "__main__":
    main()
