#!/usr/bin/env python3
"""bench.py -- generations/sec of the reproduction hot path on BASELINE.json's config 2: 100k individuals x 1M biallelic
SNPs, 1 chromosome of 100 Mb, uniform recombination map (2001 rows, 5e-4/row), mutation 1e-8/bp (5e-4/row), 1000 CVs.

    python bench.py [--gpus N --steps K --warmup W]             (N > 1: starts its own N rank processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus 2 --migration-rate 0.01              (BASELINE config 3: two populations exchanging 1 % per generation)

One process per GPU; each rank advances its OWN population of the full config-2 size (weak scaling: populations shard across
GPUs); no data-path collective is needed without migration.  A step = one generation in the reference's
order -- Simulation::random_mate (src/Simulation.cpp:2090), Simulation::reproduce (:2394), Simulation::ras_compute_AD (:2624) --
all of it on the GPU behind ONE pair of C-ABI calls (gev_generation_begin / gev_generation_end): the generation's 2 + N*nchr
ras_glob_seed() values are drawn on the device from the host's glob_generator state, the couples are formed there, the state
behind the draws comes back with the sexes and A/D.  With selection "none" (this workload) the next generation needs nothing the
host derives from this one's A/D, so the loop hands generation g+1 over before it reads generation g's A/D (published in the
library's second pinned buffer); --no-host-overlap reads first.  gev_set_generation_chain(0) tells the library that the host
draws nothing from glob_generator between two generations: it then samples generation g+1 while generation g's rows are
stitched.  --mating host is round 2's loop (numpy stand-in for random_mate, seeds drawn by a second host thread).  The founder
panel is generated on the device before the timed region, so genotype state is resident in HBM throughout.  --plane-less times
BASELINE config 5's mode (interval state only) and is NOT the headline configuration.

Prints ONE JSON line (rank 0).
  roofline      prices the dense stitch (k_stitch_segments).  A genotype row is kept as segments (2 KiB by default, GEV_SEG_CHUNKS);
                in a segment that contains none of its crossover boundaries an offspring gamete is one parental haplotype unchanged:
                that segment names the parent's unit and is not copied.  The units of a launch are the segments it WRITES (about one
                per crossover; gev_stitch_totals); algorithmic bytes per launch = bytes written x 2 (read once, written once;
                SURVEY.md 8(d)'s per-gamete figure restricted to the bytes that change hands), divided by the kernel's duration
                measured with HIP events on the library's own stream.  frac / kernel_ms = live, inside the timed region, where the
                kernel runs behind the generation's small work while the host turns the generation around; isolated_* = the same
                kernel alone (extra untimed generations with the streams serialised).  `traffic` = measured HBM bytes (newest
                committed rocprofv3 PMC passes of the same launch) or null.
  sampling_kernels   the Bernoulli-draw kernels (k_rec_sample8, k_mut_sample8 + the slow-task kernels): draws/s live and alone, and
                the share of the VALU issue peak (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave instruction, MI355X_MICROARCH.md)
                their measured VALU instruction count (committed PMC pass) amounts to.
  sustained_300 the same loop for 300 more generations after the timed region (lists and pools age): generations/s over them and
                the step time of the first and last 20.
  cli_dropin    the newest committed timing of the reference's own program bound to the library (tools/cli_timing.py; not run here).
  cpu_baseline  the unmodified reference (oracle/_ref/ref_harness, kind "reference"; when the binary is absent the bit-exact CPU
                oracle, kind "port", and `reference_binary_present` false) on a bounded sample of the same workload, 1 host thread.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(args, n_loci_unused):
    """oracle on a bounded sample: N_s individuals, same maps / CVs; the reference's (and the
    oracle's) generation cost is linear in N and independent of the SNP count,
    so generations/sec at N = 100k is rate(N_s) * N_s / N."""
    from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate
    from oracle import oracle_api
    from tests.synth import synth_packed
    ns = args.cpu_sample_ind
    cfg = SyntheticConfig(ns, 1024, seed=12345)
    ol = oracle_api.load()
    o = ol.create(1, 1, 1)
    cfg.apply_static(o)
    o.upload_founders(0, 0, synth_packed(1, 2 * ns, 1024), 1024)
    o.upload_cv_founders(0, 0, 0, synth_packed(2, 2 * ns, 1000), 1000)
    so = Simulation(o, 12345, 1, True)
    so.ras_initial_human_gen0(0, ns)
    rng = np.random.default_rng(0)
    times = []
    for gen in range(1, args.cpu_sample_gens + 1):
        so.couples[0] = synthetic_random_mate(so.sex[0], ns, rng)
        t0 = time.perf_counter()
        so.reproduce(0, gen)
        so.ras_compute_AD(0, gen)
        times.append(time.perf_counter() - t0)
    o.close()
    per_gen = float(np.mean(times))
    return {"value": (1.0 / per_gen) * ns / args.n_ind, "unit": "generations/s", "cores": 1, "kind": "port",
            "sample": f"oracle (bit-exact CPU port) timed on {ns} individuals x {args.cpu_sample_gens} generations "
                      f"({per_gen:.2f} s/generation), same maps/CVs/mutation rate; scaled linearly in N to {args.n_ind} individuals "
                      f"(generation cost is independent of the SNP count); host has {os.cpu_count()} cores, 1 used"}


def write_bits_text(path, bits_rows_by_cols):
    """rows x cols 0/1 matrix -> the reference's .hap text ('0 1 0 ...' per line) without Python loops"""
    r, c = bits_rows_by_cols.shape
    buf = np.full((r, 2 * c), ord(" "), dtype=np.uint8)
    buf[:, 0::2] = bits_rows_by_cols + ord("0")
    buf[:, -1] = ord("\n")
    buf.tofile(path)


def cpu_baseline_reference(args):
    """The REAL reference (oracle/_ref/ref_harness: unmodified GeneEvolve objects + a timing/dump driver, built
    in the build container by oracle/Makefile.ref) on a bounded sample of the same workload: wall time of its
    Simulation::reproduce + Simulation::ras_compute_AD calls, scaled linearly in N."""
    import subprocess
    import tempfile
    from geneevolve_amd.host import SyntheticConfig
    from tests.synth import synth_bits
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(exe):
        return None
    ns, gens, nsnp = args.ref_sample_ind, args.ref_sample_gens, 64
    cfg = SyntheticConfig(ns, nsnp, seed=12345)
    with tempfile.TemporaryDirectory(prefix="gev_ref_") as wd:
        j = lambda f: os.path.join(wd, f)
        write_bits_text(j("ref.hap"), np.ascontiguousarray(synth_bits(1, 2 * ns, nsnp).T))         # .hap is SNP-major
        with open(j("ref.legend"), "w") as f:
            f.write("id pos al0 al1\n" + "".join(f"rs{i+1} {int(p)} A C\n" for i, p in enumerate(cfg.snp_pos)))
        with open(j("ref.indv"), "w") as f:
            f.write("".join(f"id{i+1}\n" for i in range(ns)))
        with open(j("hapaddr.txt"), "w") as f:
            f.write("chr hap legend sample\n" + f"1 {j('ref.hap')} {j('ref.legend')} {j('ref.indv')}\n")
        cM = 1e-6 * (cfg.rmap_bp - cfg.rmap_bp[0]).astype(np.float64)                                # 1 cM/Mb -> 5e-4 per 50 kb row
        with open(j("rmap.txt"), "w") as f:
            f.write("chr bp cM\n" + "".join(f"1 {int(b)} {float(c)!r}\n" for b, c in zip(cfg.rmap_bp, cM)))
        with open(j("mmap.txt"), "w") as f:
            f.write("chr bp mutation_rate\n" + "".join(f"1 {int(b)} {float(r)!r}\n" for b, r in zip(cfg.mut_bp, cfg.mut_rate)))
        bp, a, d = cfg.cv[0][0]
        with open(j("cvinfo.txt"), "w") as f:
            f.write("chr pos a d\n" + "".join(f"1 {int(b)} {float(x)!r} {float(y)!r}\n" for b, x, y in zip(bp, a, d)))
        write_bits_text(j("cv.hap"), np.ascontiguousarray(synth_bits(2, 2 * ns, len(bp)).T))
        with open(j("cvaddr.txt"), "w") as f:
            f.write(f"1 {j('cv.hap')}\n")
        with open(j("popinfo.txt"), "w") as f:
            f.write("pop_size mat_cor offspring_dist selection_func selection_func_par1 selection_func_par2\n" + f"{ns} 0 p thr 1 1\n" * gens)
        cmd = [exe, "--file_gen_info", j("popinfo.txt"), "--file_hap_name", j("hapaddr.txt"), "--file_recom_map", j("rmap.txt"),
               "--file_mutation_map", j("mmap.txt"), "--file_cv_info", j("cvinfo.txt"), "--file_cvs", j("cvaddr.txt"),
               "--va", "0.5", "--vd", "0", "--ve", "0.5", "--RM", "--seed", "12345", "--prefix", j("out")]
        env = dict(os.environ, GEV_DUMP=j("d"), GEV_DUMP_GENS="9999")
        try:
            subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=600)
            t = np.loadtxt(j("d.timing.txt")).reshape(-1, 5)
        except Exception:  # noqa: BLE001  (binary missing its runtime, timeout ...): fall back to the port
            return None
    per_gen = float((t[:, 3] + t[:, 4]).mean())
    return {"value": (1.0 / per_gen) * ns / args.n_ind, "unit": "generations/s", "cores": 1, "kind": "reference",
            "sample": f"unmodified reference objects (oracle/_ref/ref_harness) on {ns} individuals x {gens} generations: "
                      f"Simulation::reproduce {t[:,3].mean():.2f} s + ras_compute_AD {t[:,4].mean():.2f} s per generation, same maps/CVs/mutation "
                      f"rate, --RM; scaled linearly in N to {args.n_ind} individuals (the reference's generation cost is independent "
                      f"of the SNP count); host has {os.cpu_count()} cores, 1 used (the reference is single threaded)"}


def launch_ranks(n):
    """One worker process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in its environment, the same contract as
    torch.distributed.run), started from a parent that has not touched the GPU.  Rank 0's stdout is this process's stdout.
    A failing rank ends the others (by PID) and its exit code is returned."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                        # a rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plane-less", action="store_true", help="interval state only (gev_set_dense_state 0): the mode of populations that exceed HBM; no stitch, roofline not applicable")
    ap.add_argument("--no-presample", action="store_true", help="do not hand the next generation's seeds to the library ahead of its couples (gev_presample)")
    ap.add_argument("--n-ind", type=int, default=100_000)
    ap.add_argument("--n-loci", type=int, default=1_000_000)
    ap.add_argument("--n-cv", type=int, default=1000)
    ap.add_argument("--map-step", type=int, default=50_000, help="bp between recombination / mutation map rows (config 2: 50 kb = 2001 rows; 100 = one row per locus, the sampling stress case)")
    ap.add_argument("--nchr", type=int, default=1, help="chromosomes (each of --n-loci SNPs and 100 Mb; config 2 has 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-ind", type=int, default=20000)
    ap.add_argument("--cpu-sample-gens", type=int, default=4)
    ap.add_argument("--ref-sample-ind", type=int, default=30000, help="cpu_baseline: individuals of the reference's sample run (10-20 s of its reproduce + ras_compute_AD; its cost per individual grows with the size, so a larger sample is the fairer one)")
    ap.add_argument("--ref-sample-gens", type=int, default=4)
    ap.add_argument("--no-intervals", action="store_true", help="do not keep the ancestry interval state on the device")
    ap.add_argument("--migrant-rows", action="store_true",
                    help="with --migration-rate: the migrants' genotype rows travel in the records (250 KB per migrant at config 2); default: they travel as "
                         "lists and the importing GPU rebuilds the rows from the founder panels of all populations (gev_set_migrant_rows 0)")
    ap.add_argument("--migration-rate", type=float, default=0.0,
                    help="N>1 only: fraction of each population that moves to EACH other population every generation "
                         "(BASELINE config 3 uses 0.01); rows travel by all_to_all over RCCL")
    ap.add_argument("--spawn", action="store_true", help="go through the worker launcher even for --gpus 1 (checks that the launcher costs nothing)")
    ap.add_argument("--mating", choices=["device", "host"], default="device",
                    help="device (default): Simulation::random_mate and the generation's ras_glob_seed() draws run on the GPU, a step is one "
                         "gev_generation_begin/_end pair; host: round 2's loop (numpy stand-in for random_mate, seeds drawn by a second host thread)")
    ap.add_argument("--no-host-overlap", action="store_true", help="--mating device: read a generation's A/D before the next generation is handed over (default: hand over first)")
    ap.add_argument("--no-chain", action="store_true", help="--mating device: no head start across generations (gev_set_generation_chain)")
    ap.add_argument("--no-pipeline", action="store_true", help="--mating host only: mate, then gev_reproduce, strictly one after the other (default: the host forms the next couples between gev_reproduce_begin and _end)")
    ap.add_argument("--isolated-steps", type=int, default=3, help="extra untimed generations without stream overlap for roofline.isolated")
    ap.add_argument("--sustained-steps", type=int, default=300, help="generations run after the timed region for the sustained_300 block (0: none; default config and one GPU only)")
    args = ap.parse_args()

    # `python bench.py --gpus N` as typed: no rendezvous environment yet -> this process becomes the launcher.  It starts one
    # fresh worker process per GPU BEFORE any GPU call is made here (it never imports torch) and relays rank 0's JSON line.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                               # under torch.distributed.run the launcher's world size wins
        args.gpus = world
    # rehearsal knob for a ONE-GPU box: all ranks share device 0 and talk over gloo (RCCL refuses two ranks on one GPU)
    one_gpu = os.environ.get("GEV_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from geneevolve_amd.capi import GevLibrary
    from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate
    lib = GevLibrary()                                   # the HIP library or nothing
    scale = args.map_step / 50_000.0                     # --map-step: same expected crossovers / mutations per gamete on a finer map (SURVEY.md 8(d) "C2-stress")
    cfg = SyntheticConfig(args.n_ind, args.n_loci, nchr=args.nchr, n_cv=args.n_cv, seed=12345, map_step=args.map_step,
                          rec_per_row=5e-4 * scale, mut_per_row=5e-4 * scale)   # same grids / maps / CV effects on every rank
    migrate = world > 1 and args.migration_rate > 0
    n_pop_ctx, my_pop = (world, rank) if migrate else (1, 0)   # with migration every rank knows all populations' static tables
    ctx = lib.create(n_pop_ctx, args.nchr, 1, local_rank)
    if args.no_intervals:
        ctx.set_track_intervals(False)
    if args.plane_less:                                         # BASELINE config 5's mode (not the headline): no resident genotype planes
        ctx.set_dense_state(False)
    for p in range(n_pop_ctx):
        cfg.apply_static(ctx, p)
    P = my_pop
    if migrate:
        # immigrants are appended behind the residents before the emigrants' slots are reused: room for them from the start, so
        # that the first exchange does not regrow (and copy) the planes
        ctx.reserve(P, args.n_ind + int(round(args.migration_rate * args.n_ind)) * (world - 1))
    for c in range(args.nchr):
        if not args.plane_less:
            ctx.synth_founders(P, c, 2 * args.n_ind, 1000 + 100 * c + rank)
        ctx.synth_cv_founders(P, 0, c, 2 * args.n_ind, 2000 + 100 * c + rank)
    migrant_lists = migrate and not args.migrant_rows and not args.plane_less and not args.no_intervals
    if migrant_lists:
        # the panels of ALL populations live on every GPU: n_pop x 2N rows of the plane's stride, next to this rank's two plane sets
        # and its list arenas.  Where they do not fit (config 2 on 8 GPUs: 8 x 25 GB), the rows travel.
        row_bytes = -(-args.n_loci // 1024) * 128
        panels = n_pop_ctx * args.nchr * 2 * args.n_ind * row_bytes
        planes = 2 * args.nchr * 2 * args.n_ind * row_bytes
        total = torch.cuda.get_device_properties(local_rank).total_memory
        share = 2 if one_gpu else 1                                  # (rehearsal: the ranks share one GPU)
        if (panels + planes + 0.12 * total) * share > 0.92 * total:   # (+ list arenas, unit tables, sampling records)
            migrant_lists = False
            if rank == 0:
                print(f"bench: founder panels of {n_pop_ctx} populations ({panels / 2**30:.0f} GiB) + planes ({planes / 2**30:.0f} GiB) do not fit "
                      f"the GPU's memory: the migrants' genotype rows travel in the records (--migrant-rows)", file=sys.stderr)
    if migrant_lists:
        # every GPU keeps a read-only copy of every population's founder panel (what the reference holds as Population::hap_snps)
        for p in range(n_pop_ctx):
            for c in range(args.nchr):
                ctx.synth_founder_panel(p, c, 2 * args.n_ind, 1000 + 100 * c + p)
        ctx.set_migrant_rows(False)
    sim = Simulation(ctx, 12345 + rank, args.nchr, True)
    sim.ras_initial_human_gen0(P, args.n_ind)
    if args.mating == "device" and not args.no_chain:
        # this loop makes no ras_glob_seed() draw of its own between two generations (no phenotype scaling, no migration draws): tell the
        # library, so that it can draw the next generation's seeds and sample while this generation's rows are stitched
        ctx.set_generation_chain(0)
    rng = np.random.default_rng(rank)
    total = args.warmup + args.steps
    # Simulation::ras_glob_seed(): 1 + N*nchr draws per generation (src/Simulation.cpp:2398, :2500), made by the host INSIDE every
    # step, as the reference's reproduce does: step i draws the values of generation i+1 right after handing generation i over
    # (they are pure draws of the host's stream, so they can be made -- and given to gev_presample -- before the couples exist).
    n_seeds = 1 + args.n_ind * args.nchr
    seeds = {0: sim.ras_glob_seed(n_seeds)}              # generation 0 of the warm-up only; every later one is drawn inside a step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ad_ms, mate_ms, repro_ms, mig_ms, step_ms, seed_ms = [], [], [], [], [], []
    mig_parts = {}
    if migrate:
        from geneevolve_amd.distributed import migrate_all_to_all

    fused = args.mating == "device"
    overlap_host = fused and not args.no_host_overlap
    presample = not args.no_presample and not fused
    from concurrent.futures import ThreadPoolExecutor
    seed_pool = ThreadPoolExecutor(1)                    # the host's second thread: draws seeds while the first waits for the GPU

    def timed_draw():
        ts = time.perf_counter()
        v = sim.ras_glob_seed(n_seeds)
        return v, (time.perf_counter() - ts) * 1e3

    # Pipelined host loop (default): the sexes of a generation come out of the sampling kernels' rand() chain, which the head start
    # (gev_presample) has already run when the generation is handed over, and random mating reads nothing else -- so the host forms
    # the NEXT generation's couples between gev_reproduce_begin and gev_reproduce_end, while the device builds lists, CV planes,
    # genotype rows and A/D of this one.  --no-pipeline: mate, then gev_reproduce, one after the other.
    if migrate and not fused:
        raise SystemExit("--migration-rate needs --mating device (the library mates on the sexes that travel with the migrants)")
    pipeline = presample and not migrate and not args.no_pipeline
    state = {}
    call_ms = {}                                               # host time inside each library call of the pipelined loop, summed

    def step_pipelined(i):
        t0 = time.perf_counter()
        if "couples" not in state:                               # first step: the head start and the first couples
            sim.presample(P, seeds[i], args.n_ind)
            state["couples"] = synthetic_random_mate(sim.sex[P], args.n_ind, rng)
            seeds[i + 1] = sim.ras_glob_seed(n_seeds)
        sd = seeds.pop(i)
        ta = time.perf_counter()
        sex_new = ctx.presample_sex(P, args.n_ind)              # this generation's sexes: its sampling ran during the previous step
        tb = time.perf_counter()
        ctx.reproduce_begin(P, state["couples"], int(sd[0]), sd[1:], n_people=args.n_ind)   # Simulation::reproduce, first half
        tc = time.perf_counter()
        sim.presample(P, seeds[i + 1], args.n_ind)               # head start of the next generation, queued behind this one's lists and A/D
        td = time.perf_counter()
        fut = seed_pool.submit(timed_draw)                       # one generation's worth of ras_glob_seed() draws per step, second host thread
        t1 = time.perf_counter()
        for k, v in (("gev_presample_sex", tb - ta), ("gev_reproduce_begin", tc - tb), ("gev_presample", td - tc)):
            call_ms[k] = call_ms.get(k, 0.0) + v * 1e3
        state["couples"] = synthetic_random_mate(sex_new, args.n_ind, rng, out=state["couples"])    # host mating of the NEXT generation
        t2 = time.perf_counter()
        ctx.reproduce_end(want_sex=False)                        # ... second half (the sexes are sex_new)
        call_ms["gev_reproduce_end"] = call_ms.get("gev_reproduce_end", 0.0) + (time.perf_counter() - t2) * 1e3
        sim.sex[P] = sex_new
        sim.last_seed_reproduce = int(sd[0])
        t3 = time.perf_counter()
        sim.ras_compute_AD(P, i + 1)                             # Simulation::ras_compute_AD
        seeds[i + 2], dms = fut.result()
        seed_ms.append(dms)
        t4 = time.perf_counter()
        mate_ms.append((t2 - t1) * 1e3); repro_ms.append((t1 - t0 + t3 - t2) * 1e3); ad_ms.append((t4 - t3) * 1e3); step_ms.append((t4 - t0) * 1e3)

    def do_migration(t3):
        """Simulation::ras_do_migration: host picks WHO (:921-922), rows -- with their sexes -- go by all_to_all"""
        k = int(round(args.migration_rate * args.n_ind))
        sample = np.sort(rng.choice(args.n_ind, size=k * (world - 1), replace=False))[::-1].astype(np.uint64)
        outgoing, o = [], 0
        for j in range(world):
            outgoing.append(np.empty(0, dtype=np.uint64) if j == rank else sample[o:o + k])
            o += 0 if j == rank else k
        tr = []
        migrate_all_to_all(ctx, outgoing, P, device=f"cuda:{local_rank}", trace=tr)
        for name, sec in zip(tr[0::2], tr[1::2]) if all(isinstance(x, float) for x in tr[1::2]) else []:
            mig_parts[name] = mig_parts.get(name, 0.0) + sec * 1e3
        mig_ms.append((time.perf_counter() - t3) * 1e3)

    def step_fused(i):
        """one generation = ONE pair of library calls: random_mate (:2090) -> reproduce (:2394) -> ras_compute_AD (:2624) in the
        reference's order, its 2 + N*nchr ras_glob_seed() values drawn on the device from the host's engine state"""
        t0 = time.perf_counter()
        tb = 0.0
        if not state.get("begun"):
            ctx.generation_begin(P, sim.glob.x, args.n_ind, None)
            tb += time.perf_counter() - t0
        t1 = time.perf_counter()
        r = ctx.generation_end(want_couples=False, want_sex=True)
        t2 = time.perf_counter()
        sim.glob.x = int(r["glob_state"]); sim.sex[P] = r["sex"]; sim.last_seed_reproduce = int(r["seed_reproduce"])
        state["begun"] = False
        if overlap_host and not migrate:
            # no selection: the next generation needs nothing the host derives from this one's A/D, so it is handed over at once and
            # this generation's A/D (published, in the library's other pinned buffer) is read while the device already works
            tb0 = time.perf_counter()
            ctx.generation_begin(P, sim.glob.x, args.n_ind, None)
            tb += time.perf_counter() - tb0
            state["begun"] = True
        ta = time.perf_counter()
        sim.ras_compute_AD(P, i + 1)                             # the generation's A/D (computed inside the generation, copied out here)
        t3 = time.perf_counter()
        if migrate:
            do_migration(t3)
        call_ms["gev_generation_begin"] = call_ms.get("gev_generation_begin", 0.0) + tb * 1e3
        call_ms["gev_generation_end"] = call_ms.get("gev_generation_end", 0.0) + (t2 - t1) * 1e3
        t2 = ta
        mate_ms.append(0.0); seed_ms.append(0.0); repro_ms.append((t2 - t0) * 1e3); ad_ms.append((t3 - t2) * 1e3); step_ms.append((time.perf_counter() - t0) * 1e3)

    def step(i):
        if fused:
            return step_fused(i)
        if pipeline:
            return step_pipelined(i)
        t0 = time.perf_counter()
        sim.couples[P] = synthetic_random_mate(sim.sex[P], args.n_ind, rng, out=sim.couples.get(P))   # host mating (outside the hot path)
        t1 = time.perf_counter()
        # the next generation's ras_glob_seed() draws (inside the clock): made by a second host thread while this one waits inside
        # gev_reproduce for the GPU's small kernels, so the values are ready the moment the call returns
        fut = seed_pool.submit(timed_draw)
        sim.reproduce(P, i + 1, seeds=seeds.pop(i), n_people=args.n_ind)          # Simulation::reproduce
        seeds[i + 1], dms = fut.result()
        seed_ms.append(dms)
        if presample:                                                            # ... handed over at once: the GPU samples while the host mates
            sim.presample(P, seeds[i + 1], args.n_ind)
        t2 = time.perf_counter()
        sim.ras_compute_AD(P, i + 1)                                             # Simulation::ras_compute_AD
        t3 = time.perf_counter()
        if migrate:
            do_migration(t3)
            sim.sex[P] = None                                                    # (after a migration the host of this loop would need the migrants' sexes: --mating host does not combine with --migration-rate)
        mate_ms.append((t1 - t0) * 1e3); repro_ms.append((t2 - t1) * 1e3); ad_ms.append((t3 - t2) * 1e3); step_ms.append((time.perf_counter() - t0) * 1e3)

    # The host loop allocates a few numpy arrays per generation; CPython's cyclic collector would eventually run a FULL collection
    # over everything torch and numpy have imported (hundreds of thousands of objects: 40-60 ms, once, about 300 generations into
    # the loop -- it showed up as one 45 ms step in `sustained_300`).  What exists now is long-lived: move it out of the collector's way.
    import gc
    gc.collect(); gc.freeze()
    for i in range(args.warmup):
        step(i)
    if state.get("begun"):                               # (timing_totals below needs an idle context)
        r = ctx.generation_end(want_couples=False, want_sex=True)
        sim.glob.x = int(r["glob_state"]); sim.sex[P] = r["sex"]; state["begun"] = False
    del ad_ms[:], mate_ms[:], repro_ms[:], mig_ms[:], step_ms[:], seed_ms[:]
    mig_parts.clear(); call_ms.clear()
    tot0, n0 = ctx.timing_totals()                       # (implies a sync of both library streams)
    rows0 = (0, 0, 0, 0) if args.plane_less else ctx.stitch_totals()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    barrier()                                            # torch.cuda.synchronize() waits for the last dense stitch too (and for the generation handed over ahead, if any)
    dt = time.perf_counter() - t0
    rows1 = (0, 0, 0, 0) if args.plane_less else ctx.stitch_totals()
    if state.get("begun"):                               # the generation handed over in the last step: collected outside the clock (every timed step collected one)
        r = ctx.generation_end(want_couples=False, want_sex=True)
        sim.glob.x = int(r["glob_state"]); sim.sex[P] = r["sex"]; state["begun"] = False
    tot1, n1 = ctx.timing_totals()
    assert n1 - n0 in (args.steps, args.steps + 1)
    ng = n1 - n0
    sample_ms = [(tot1[0] - tot0[0]) / ng]; stitch_ms = [(tot1[1] - tot0[1]) / ng]; sparse_ms = [(tot1[2] - tot0[2]) / ng]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # host-side means of the timed region (the sustained run below appends to the same lists)
    host_means = {"host_mating": float(np.mean(mate_ms)), "host_seed_draws": float(np.mean(seed_ms)), "gev_reproduce_wall": float(np.mean(repro_ms)),
                  "gev_compute_ad_wall": float(np.mean(ad_ms)), "migration_wall": float(np.mean(mig_ms)) if mig_ms else None,
                  "migration_steps": {k: v / args.steps for k, v in mig_parts.items()} if mig_parts else None,
                  "host_ms_inside_calls": {k: v / args.steps for k, v in call_ms.items()} if call_ms else None}
    main_step_ms = [round(x, 2) for x in step_ms[:args.steps]]
    # the same loop for a few hundred generations more: lists lengthen, pools and arenas age (not part of `value`)
    sustained = None
    if args.sustained_steps > 0 and world == 1 and fused and not args.plane_less:
        del step_ms[:]
        barrier()
        ts = time.perf_counter()
        for i in range(total, total + args.sustained_steps):
            step(i)
        barrier()
        dts = time.perf_counter() - ts
        if state.get("begun"):
            r = ctx.generation_end(want_couples=False, want_sex=True)
            sim.glob.x = int(r["glob_state"]); sim.sex[P] = r["sex"]; state["begun"] = False
        med = float(np.median(step_ms)); slow = [(total + 1 + i, round(v, 2)) for i, v in enumerate(step_ms) if v > 2 * med]
        k20 = min(20, len(step_ms))
        first, last = float(np.mean(step_ms[:k20])), float(np.mean(step_ms[-k20:]))
        sustained = {"generations": args.sustained_steps, "from_generation": total + 1, "generations_per_s": args.sustained_steps / dts, "ms_per_step": dts / args.sustained_steps * 1e3,
                     "host_step_ms_first_20": first, "host_step_ms_last_20": last, "last_over_first": last / first if first > 0 else None,
                     "host_step_ms_median": med, "steps_over_twice_the_median": slow[:32], "ms_in_those_steps": float(sum(v for _, v in slow)),
                     "redone_generations": int(ctx.redo_count()), "list_pieces": ctx.list_stats(P, 0)}
        total += args.sustained_steps
        tot1s, n1s = ctx.timing_totals()

    # a few extra, untimed generations with the two library streams serialised: the stitch kernel alone on the GPU
    iso = None; iso_sampling = None
    if args.isolated_steps > 0 and not migrate and not args.plane_less:
        ctx.set_overlap(False)
        ta, na = ctx.timing_totals()
        for j in range(args.isolated_steps):
            if fused:
                sim.next_generation_rm(P, args.n_ind)
                continue
            sim.couples[P] = synthetic_random_mate(sim.sex[P], args.n_ind, rng, out=sim.couples.get(P))
            sim.reproduce(P, total + j + 1, seeds=seeds.pop(total + j) if (total + j) in seeds else sim.ras_glob_seed(n_seeds), n_people=args.n_ind)
        tb, nb = ctx.timing_totals()
        iso = (tb[1] - ta[1]) / max(nb - na, 1)
        iso_sampling = (tb[0] - ta[0]) / max(nb - na, 1)
        ctx.set_overlap(True)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        gens_per_s = world * args.steps / dt
        # Units one stitch launch processes = the gametes it copies.  A gamete without a crossover is its parent's haplotype
        # unchanged (Simulation::recombine, src/Simulation.cpp:2910): its slot points at the parent's row and nothing is copied,
        # so it is not a unit of the launch.  Per unit: L/8 bytes read + L/8 written.
        full_bytes = args.n_ind * args.n_loci / 2.0 * args.nchr   # every gamete copied: N*L/2 per generation (the figure of earlier rounds)
        bytes_written, bytes_total = rows1[0] - rows0[0], rows1[1] - rows0[1]
        segs_written, segs_total = rows1[2] - rows0[2], rows1[3] - rows0[3]
        written_frac = bytes_written / bytes_total if bytes_total else 1.0
        alg_bytes = 2.0 * bytes_written / max(args.steps, 1)         # per generation: every written byte is read once and written once
        stitch = float(np.mean(stitch_ms))
        achieved = alg_bytes / (max(stitch, 1e-9) * 1e-3) / 1e9 if not args.plane_less else 0.0
        full_equiv = full_bytes / (max(stitch, 1e-9) * 1e-3) / 1e9 if not args.plane_less else 0.0
        traffic, traffic_src = None, None                   # HBM bytes per launch from the committed PMC passes (same workload only)
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_config2_pmc_hbm.json")))       # newest round last
        if pmcs and (args.n_ind, args.n_loci, args.nchr, args.map_step) == (100_000, 1_000_000, 1, 50_000) and not args.plane_less:
            # newest committed measurement of the SAME launch: its algorithmic bytes must be this run's (a measurement of the
            # every-gamete-copied kernel does not describe the shared-row one, and vice versa)
            for f in reversed(pmcs):
                sm = json.load(open(f)).get("stitch_summary")
                if sm and abs(sm["algorithmic_bytes_per_launch"] - alg_bytes) <= 0.02 * alg_bytes:
                    traffic, traffic_src = sm["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
                    break
        # what the memory system really moved per second while the kernel ran: measured bytes / measured time.  The kernel reads a
        # parent chunk once for all of its gametes, so this is BELOW `achieved` (which prices the algorithmic N*L/2 bytes).
        hbm_actual = traffic / (max(stitch, 1e-9) * 1e-3) / 1e9 if traffic else None
        out = {
            "metric": "generations/sec", "value": gens_per_s,
            "unit": f"generations/s of {args.n_ind} individuals x {args.n_loci} loci populations (summed over GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 bit-planes + f64 A/D",
            "data": "synthetic (device-generated founder panel, uniform maps, SURVEY.md 8(d) config C2)",
            "config": {"workload": ("BASELINE config 2: 100k individuals x 1M SNPs, 1 population per GPU, 1 chromosome (100 Mb), "
                                    "uniform recombination map 2001 rows @ 5e-4, mutation 1e-8/bp, 1000 CVs, random mating")
                       if (args.n_ind, args.n_loci, args.nchr, args.map_step) == (100_000, 1_000_000, 1, 50_000) else
                       f"config-2 family, non-default size: {args.n_ind} individuals x {args.nchr} chromosome(s) x {args.n_loci} SNPs, map rows every {args.map_step} bp",
                       "n_individuals": args.n_ind, "n_loci": args.n_loci, "n_chromosomes": args.nchr, "n_cv": args.n_cv, "parallelism": f"{world} population(s), 1 per GPU", "migration_rate": args.migration_rate if migrate else 0.0,
                       "migrants_travel_as": ("lists (rows rebuilt from founder panels)" if migrant_lists else "whole genotype rows") if migrate else None,
                       "interval_state_tracked": not args.no_intervals, "resident_genotype_planes": not args.plane_less,
                       "mating": "reference Simulation::random_mate on the device (gev_generation_begin: random_mate -> reproduce -> ras_compute_AD per step, in the reference's order)" if fused
                                 else "host, numpy stand-in for random_mate (round 2's loop)",
                       "selection_function": "none (selection_value_func = 1 for everyone)",
                       "next_generation_handed_over_before_this_ones_A_D_is_read": bool(overlap_host and not migrate),
                       "seeds_handed_over_before_couples": presample,
                       "host_mating_overlaps_device_work": pipeline,
                       "ras_glob_seed_draws": "on the device, from the host's glob_generator state (2 + N*nchr per generation)" if fused else
                                              "inside the timed loop, one generation's worth per step, by a second host thread (phase_ms.host_seed_draws)"},
            "loci_individuals_per_sec": world * args.steps * args.n_ind * args.n_loci * args.nchr / dt,
            "phase_ms": dict({"sampling": float(np.mean(sample_ms)), "sparse_lists_and_cv_planes": float(np.mean(sparse_ms)), "dense_stitch": stitch}, **host_means),
            "host_step_ms": main_step_ms,
            "sustained_300": sustained,
            "list_pieces": ctx.list_stats(P, 0) if lib.exports("list_stats") else None,
            "roofline": {"bound": "hbm", "kernel": "k_stitch_segments", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_actual_GBps": hbm_actual, "hbm_actual_frac": hbm_actual / HBM_PEAK_GBPS if hbm_actual else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": stitch,
                         "segments_written_per_launch": segs_written / max(args.steps, 1), "segments_per_launch": segs_total / max(args.steps, 1),
                         "written_fraction_of_row_bytes": written_frac,
                         "every_gamete_copied_equivalent_GBps": full_equiv, "every_gamete_copied_equivalent_frac": full_equiv / HBM_PEAK_GBPS,
                         "segment_bytes": (bytes_total / segs_total) if segs_total else None,
                         "note": "units of a launch = the row segments it writes (segment_bytes each on average; those that contain a crossover boundary -- every other segment "
                                 "of an offspring row names the parental unit and is not copied); algorithmic bytes = bytes written x 2 (each is read once "
                                 "and written once); every_gamete_copied_equivalent_* prices all 2N whole rows (N*L/2 bytes, the definition of round 1) "
                                 "and can exceed the peak; kernel time measured live with HIP events on the library's stitch stream inside the timed region (the stitch is "
                                 "enqueued behind the generation's small work and runs while the host turns the generation around; whatever else is on the GPU "
                                 "then -- the next generation's first kernels, its sampling -- shares it); isolated_* = same kernel with the streams "
                                 "serialised (extra untimed generations); traffic = rocprofv3 PMC measurement committed under profiles/",
                         "isolated_kernel_ms": iso, "isolated_achieved": (alg_bytes / (iso * 1e-3) / 1e9) if iso else None,
                         "isolated_frac": (alg_bytes / (iso * 1e-3) / 1e9 / HBM_PEAK_GBPS) if iso else None,
                         "isolated_hbm_actual_GBps": (traffic / (iso * 1e-3) / 1e9) if (iso and traffic) else None},
        }
        # K1/K2: the sampling kernels are ALU-bound integer work (one minstd step + one threshold compare per map row and gamete)
        rows_rec = (100_000_000 // args.map_step) + 1
        draws = float(args.n_ind) * args.nchr * (2.0 * rows_rec + (rows_rec - 1))      # 2N gametes x recombination rows + N tasks x mutation rows
        samp_live = float(np.mean(sample_ms))
        VALU_PEAK = 256 * 4 * 2.4e9 / 2.0                        # wave instructions/s: 256 CUs x 4 SIMDs, one VALU instruction per wave every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
        valu, valu_src = None, None
        for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sampling_pmc_instructions.json")))):
            try:
                pm = json.load(open(f))
                valu = sum(v["SQ_INSTS_VALU"] for k, v in pm.items() if isinstance(v, dict) and "SQ_INSTS_VALU" in v)
                valu_src = os.path.relpath(f, ROOT)
                break
            except Exception:  # noqa: BLE001
                continue
        default_shape = (args.n_ind, args.n_loci, args.nchr, args.map_step) == (100_000, 1_000_000, 1, 50_000)
        out["sampling_kernels"] = {
            "kernels": "k_mut_sample8 + k_mut_sample + k_rec_sample8 + k_rec_sample (+ the ras_glob_seed draws of the generation)",
            "bernoulli_draws_per_generation": draws, "live_ms": samp_live, "isolated_ms": iso_sampling,
            "draws_per_s_live": draws / (samp_live * 1e-3) if samp_live > 0 else None,
            "draws_per_s_isolated": draws / (iso_sampling * 1e-3) if iso_sampling else None,
            "valu_wave_instructions_per_generation": valu if default_shape else None, "valu_source": valu_src if default_shape else None,
            "valu_issue_peak_wave_instructions_per_s": VALU_PEAK,
            "frac_of_valu_issue_peak_isolated": (valu / (iso_sampling * 1e-3) / VALU_PEAK) if (valu and iso_sampling and default_shape) else None,
            "frac_of_valu_issue_peak_live": (valu / (samp_live * 1e-3) / VALU_PEAK) if (valu and samp_live > 0 and default_shape) else None,
            "note": "bound: integer VALU issue (no HBM traffic to speak of: the map thresholds sit in LDS); live = next to the dense stitch and the A/D chain, where the "
                    "kernels get a share of the SIMDs; instruction count from the committed PMC pass (SQ_INSTS_VALU summed over the sampling kernels of one generation)"}
        clis = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cli_dropin.json")))
        if clis:
            try:
                out["cli_dropin"] = dict(json.load(open(clis[-1])), source=os.path.relpath(clis[-1], ROOT),
                                         note="the reference's own command-line program with its hot path bound to the library (integration/), timed by tools/cli_timing.py on one MI355X box; not measured by this run")
            except Exception:  # noqa: BLE001
                out["cli_dropin"] = None
        else:
            out["cli_dropin"] = None
        if not args.no_cpu_baseline and world == 1:          # CPU baseline: rank 0 at N=1 only
            port = cpu_baseline(args, args.n_loci)
            ref = cpu_baseline_reference(args)
            out["cpu_baseline"] = dict(ref if ref is not None else port, reference_binary_present=ref is not None)
            if ref is None:
                print("bench.py: oracle/_ref/ref_harness is absent or failed: cpu_baseline is the CPU oracle (kind \"port\"), not the reference itself", file=sys.stderr)
            out["cpu_baseline_port"] = port
            out["gpu_over_cpu"] = (gens_per_s / world) / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
