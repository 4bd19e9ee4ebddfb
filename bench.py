#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s on uniform-random u32 keys (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete radix sort of one batch of synthetic keys that is already resident in HBM:
  N = 1 : BASELINE config #2, key-only RadixSort32, 64Mi uniform-random u32 keys, in place
          (Pprims::radixSort path: adlhip_radix_sort_u32).
  N > 1 : every rank holds 64Mi keys (weak scaling); a step = top-byte partition -> balanced splitters from the
          all-reduced histogram -> RCCL all-to-all -> local sort (oclradixsort_amd/dist.py).  The K steps are
          independent batches fed through ShardedRadixSort.sort_stream: the exchange of batch i+1 (xGMI-bound)
          overlaps the local sort of batch i (HBM-bound); the un-pipelined per-step time and stage breakdown are
          reported under "serial" beside the metric.  Beside it, under "config4", BASELINE config #4 at its stated
          size: 2^30 keys in total, 2^30 / N per GPU (2^27 at N = 8), a few steps through the same pipeline.
Every step sorts its OWN pre-generated random buffer (K + W buffers of 256 MiB are generated on the
device before the timed region), so no step sees pre-sorted data and no restore copy is timed.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     : dominant kernel = the launch kind of a sort that takes the most time (pass 1 of the large sort at N = 1),
                 ALGORITHMIC bytes per launch (what that sweep must read and write; SURVEY.md section 8d's n*E read +
                 n*E write, less where a slab holds 16-bit keys) / its average launch duration, measured here with
                 hipEvents on the library's own stream in a second, profiled loop over the same inputs
                 (toggleProfiling brackets every launch with an event pair).  roofline.by_kernel lists every sweep of
                 the sort the same way (pass 1 / pass 2 / finish); roofline.traffic = HBM bytes per launch of the
                 dominant kernel from FETCH_SIZE / WRITE_SIZE, collected by two child runs of this file under
                 `rocprofv3 --kernel-trace --pmc ...` (separate passes, kernel trace only) when rocprofv3 is there.
  cpu_baseline : the reference's single-threaded CPU sort (oracle/_ref, else the oracle port) timed on
                 this host on the same 64Mi-key workload (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_KEYS = 1 << 26            # 64Mi keys per GPU
ELEM_BYTES = 4
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
CONFIG4_TOTAL_KEYS = 1 << 30     # BASELINE config #4: 1B u32 keys over the GPUs of one node


def cpu_baseline(n):
    """Reference CPU path (single thread) on a bounded sample: the same n-key workload, repeated until
    about 10 s of CPU work is reached (at least 2, at most 4 runs)."""
    import oracle
    keys = oracle.keys_u32(n, seed=123)
    kind = "reference" if oracle.have_ref() else "port"
    fn = oracle.ref_sort_u32 if kind == "reference" else oracle.sort_u32
    times = []
    t_total = 0.0
    while len(times) < 2 or (t_total < 10.0 and len(times) < 4):
        work = keys.copy()
        t0 = time.perf_counter()
        if kind == "reference":
            oracle.ref().ref_radix_sort_u32(work.ctypes.data_as(oracle._u32p), work.size)
        else:
            oracle.lib().oracle_radix_sort_u32(work.ctypes.data_as(oracle._u32p), work.size)
        dt = time.perf_counter() - t0
        times.append(dt)
        t_total += dt
    best = min(times)
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n / best / 1e9, "unit": "Gkeys/s", "cores": 1, "kind": kind,
            "sample": "%d runs of the full %d-key u32 workload, best run %.3f s; host: %s, %d logical cores, 1 used "
                      "(the reference's CPU sort is single-threaded)"
                      % (len(times), n, best, cpu_model, os.cpu_count() or 0)}, fn


# rocprofv3 kernel names of the launch kinds bench.py prices (csrc/adlhip.hip launch names -> substrings of the kernel's name)
SETTLE_SORTS = 80   # ~35 ms of sorting before the warm-up steps (see main)
PMC_KERNEL = {
    "msd2_pass1_u32": "msd_scatter_persist_kernel<unsigned int, 512, 32, 1,",
    "msd2_pass2_u32": "msd_scatter_persist_kernel<unsigned int, 512, 32, 2,",
    "segment_sort_wave_u32": "wave_finish16_kernel<",
    "onesweep_u32_8b": "onesweep_chain_kernel",
    # configs #3 and #5 (both look-back passes of a sort are one kernel; the counters are averaged over the two)
    "msd2s_pass1_kv32": "msd_lookback_scatter_kernel<unsigned long, 512, 16, false>",
    "msd2s_pass2_kv32": "msd_lookback_scatter_kernel<unsigned long, 512, 16, false>",
    "segment_sort_wave_e64": "wave_segment_sort_kernel<unsigned long, 24,",
    "msd2s_pass1_u64": "msd_lookback_scatter_kernel<unsigned long, 512, 16, true>",
    "msd2s_pass2_u64": "msd_lookback_scatter_kernel<unsigned long, 512, 16, true>",
    "segment_sort_bin_u64": "bin_segment_sort_kernel<unsigned long",
}


def pmc_child(n, steps):
    """What a `rocprofv3 --pmc` pass of this file runs: the sorts and nothing else."""
    from oclradixsort_amd import Buffer, DeviceUtils, Pprims
    d = DeviceUtils.allocate()
    p = Pprims()
    bufs = [Buffer(d, n, np.uint32) for _ in range(steps + 1)]
    for i, b in enumerate(bufs):
        b.generate(n, seed=123 + i)
    for b in bufs:
        p.radixSort(d, b, n)
    DeviceUtils.waitForCompletion(d)
    for b in bufs:
        b.release()
    if n == N_KEYS:   # configs #3 and #5: two sorts each
        for nn, gen, fn in ((1 << 26, 1, p.radixSort), (1 << 28, 2, p.radixSort64)):
            bb = [Buffer(d, nn, np.uint64) for _ in range(2)]
            for i, b in enumerate(bb):
                b.generate(nn, seed=900 + i, kind=gen)
            for b in bb:
                fn(d, b, nn)
            DeviceUtils.waitForCompletion(d)
            for b in bb:
                b.release()
    p.close()
    DeviceUtils.deallocate(d)


def measure_traffic(n, launch_names):
    """HBM bytes per launch of the named launch kinds: FETCH_SIZE and WRITE_SIZE in their own rocprofv3 passes (kernel trace only),
    corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (both in KiB; FETCH_SIZE counts half of the bytes of a
    streaming read -> x2).  Returns {launch name: {...}} or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    res = {k: {} for k in launch_names}
    tmp = tempfile.mkdtemp(prefix="adlhip_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--n", str(n), "--steps", "3"]
            # a process group of its own: on a timeout the profiler AND the Python child under it are killed -- a survivor would
            # keep its buffers on the GPU and run beside the timed region
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=300)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                raise
            if rc != 0:
                raise RuntimeError("rocprofv3 %s pass failed with code %d" % (ctr, rc))
            acc = {k: [0.0, 0] for k in launch_names}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != ctr:
                            continue
                        for k in launch_names:
                            if PMC_KERNEL[k] in row.get("Kernel_Name", ""):
                                acc[k][0] += float(row["Counter_Value"])
                                acc[k][1] += 1
            for k in launch_names:
                if acc[k][1]:   # (kernels of paths the sort did not take have no rows)
                    res[k][ctr + "_kb"] = acc[k][0] / acc[k][1]
                    res[k]["launches_averaged"] = acc[k][1]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k in launch_names:
        if "FETCH_SIZE_kb" in res[k] and "WRITE_SIZE_kb" in res[k]:
            res[k]["traffic_bytes_per_launch"] = int(res[k]["FETCH_SIZE_kb"] * 1024 * 2.0 + res[k]["WRITE_SIZE_kb"] * 1024)
            out[k] = res[k]
    if not out:
        raise RuntimeError("no counter rows for any of the sort's kernels")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--keys-per-gpu", dest="n", type=int, default=int(os.environ.get("ADLHIP_BENCH_N", N_KEYS)),
                    help="keys per GPU (default 64Mi = BASELINE config #2; under torch.distributed.run use --keys-per-gpu "
                         "or ADLHIP_BENCH_N: its own parser claims --n)")
    ap.add_argument("--algo", type=int, default=None, help="sort.algo override (0 onesweep, 1 three-kernel)")
    ap.add_argument("--digit-bits", type=int, default=None, help="sort.digit_bits override (8 or 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config #3 / #5 extras beside the metric")
    ap.add_argument("--no-settle", action="store_true", help="skip the clock-settle sorts before the warm-up steps (A/B)")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the two rocprofv3 --pmc child runs for roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.n, args.steps)
        return

    # roofline.traffic: the two rocprofv3 --pmc child runs come FIRST, before this process has touched the GPU (a process that
    # has initialised the GPU must not be the one that starts other programs on this pool), and never from inside a profiler
    pmc_traffic = {}
    under_profiler = any(("rocprof" in os.environ.get(k, "").lower()) for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) \
        or any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ)
    if (not args.no_pmc and not under_profiler and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
            and args.n == N_KEYS and os.environ.get("ADLHIP_BENCH_FORCE_DIST") != "1"):
        try:
            pmc_traffic = measure_traffic(args.n, list(PMC_KERNEL))
        except Exception as e:
            pmc_traffic = {"error": repr(e)[:200]}

    import torch
    import torch.distributed as dist
    from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
    from oclradixsort_amd.adl import Config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the device path has no CPU fallback")
    # ADLHIP_BENCH_REHEARSE=1 (development only, never a benchmark number): run the N > 1 code path with all ranks on GPU 0
    # and the collectives staged through the host over gloo -- RCCL refuses two ranks on one device, and a 1-GPU box is
    # all there is to rehearse on.  Everything else (partition, splits, streams, slots, verification) is the real thing.
    rehearse = os.environ.get("ADLHIP_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
            real_ag, real_a2a = dist.all_gather_into_tensor, dist.all_to_all_single

            def staged_all_gather(out, inp, group=None):
                torch.cuda.current_stream().synchronize()
                o = torch.empty(out.shape, dtype=out.dtype)
                real_ag(o, inp.cpu(), group=group)
                out.copy_(o)

            def staged_all_to_all(out, inp, out_splits, in_splits, group=None):
                torch.cuda.current_stream().synchronize()
                o = torch.empty(out.shape, dtype=out.dtype)
                real_a2a(o, inp.cpu(), out_splits, in_splits, group=group)
                out.copy_(o)

            dist.all_gather_into_tensor, dist.all_to_all_single = staged_all_gather, staged_all_to_all
        else:
            # RCCL's kernels share the CUs with the local sort of the previous batch: let them in first (the exchange is the
            # stage with the scarcer resource, the xGMI links)
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.n
    K, W = args.steps, args.warmup

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    # ADLHIP_BENCH_FORCE_DIST=1: take the multi-GPU code path (partition + RCCL collectives) even with one rank,
    # to exercise it on a 1-GPU box; the number it prints is NOT the N = 1 benchmark
    force_dist = os.environ.get("ADLHIP_BENCH_FORCE_DIST") == "1"
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world == 1 and not force_dist:
        d = DeviceUtils.allocate(cfg=Config(local_rank))
        if args.algo is not None:
            d.setParam("sort.algo", args.algo)
        if args.digit_bits is not None:
            d.setParam("sort.digit_bits", args.digit_bits)
        p = Pprims()
        bufs = []
        for i in range(K + W):
            b = Buffer(d, n, np.uint32)
            b.generate(n, seed=123 + i)
            bufs.append(b)
        DeviceUtils.waitForCompletion(d)
        # Clock settle (not a warm-up STEP: it sorts a buffer of its own, fresh random keys every time): the chip comes out of the
        # set-up above at a low clock and takes ~10 ms of sustained load to reach the one it holds in steady state -- the LDS-bound
        # finish of a sort runs 122 us for the first dozen sorts after an idle gap and 102 us from the thirtieth on
        # (profiles/r4_clock_ramp.txt; the HBM-bound passes do not change).  W warm-up steps of 0.35 ms each end inside that ramp.
        settle_sorts = 0 if args.no_settle else SETTLE_SORTS
        sb = Buffer(d, n, np.uint32) if settle_sorts else None

        def settle():
            for i in range(settle_sorts):
                sb.generate(n, seed=7000 + i)
                p.radixSort(d, sb, n)

        settle()
        for i in range(W):
            p.radixSort(d, bufs[i], n)
        DeviceUtils.waitForCompletion(d)
        barrier()
        sw = Stopwatch(d)
        t0 = time.perf_counter()
        sw.start()
        for i in range(W, W + K):
            p.radixSort(d, bufs[i], n)
        sw.stop()
        DeviceUtils.waitForCompletion(d)
        barrier()
        wall = time.perf_counter() - t0
        ev_ms = sw.getMs()

        # verification outside the timed region: the first timed batch vs the oracle (bit-exact)
        verified = None
        if not args.no_verify:
            import oracle
            got = bufs[W].toHost()
            want = oracle.sort_u32(oracle.keys_u32(n, seed=123 + W))
            verified = bool(np.array_equal(got, want))
            if not verified:
                raise SystemExit("bench: sorted output differs from the oracle")

        # per-kernel durations: profiled loop over fresh random inputs (same seeds regenerated in place)
        for i in range(W, W + K):
            bufs[i].generate(n, seed=123 + i)
        DeviceUtils.waitForCompletion(d)
        settle()   # the verification above left the chip idle for seconds
        d.toggleProfiling(True)
        d.profile(reset=True)
        for i in range(W, W + K):
            p.radixSort(d, bufs[i], n)
        prof = d.profile(reset=True)
        d.toggleProfiling(False)
        if sb is not None:
            sb.release()

        # Empirical ceilings on this box, measured on buffers that are COLD in every cache (the K + W key buffers in turns: 6.7 GB,
        # 25 x the Infinity Cache -- a probe that copies the same 256 MiB over and over reads its source out of that cache and reports
        # what no sweep of a sort can reach).  Every variant: plain / non-temporal loads / non-temporal stores / both, 8 workgroups
        # of 256 threads per CU.  copy = read + written bytes per second; the best variant is the ceiling rooflines are also quoted
        # against (`frac_of_copy`).
        probe = {}
        lib = __import__("oclradixsort_amd._lib", fromlist=["load"]).load()
        sink = Buffer(d, 2, np.uint64)
        sink.clear()
        nb = len(bufs)

        def rate(fn, nbytes, reps=12):
            for r in range(3):
                fn(r)
            s2 = Stopwatch(d)
            s2.start()
            for r in range(reps):
                fn(3 + r)
            s2.stop()
            return nbytes * reps / (s2.getMs() * 1e-3) / 1e9

        if nb >= 4:
            for hints, tag in ((0, "plain"), (1, "nt_loads"), (2, "nt_stores"), (3, "nt_both")):
                probe["copy_%s_GBps" % tag] = rate(lambda r, h=hints: lib.adlhip_probe_copy_ex(
                    d._h, bufs[(2 * r + 1) % nb].ptr(), bufs[(2 * r) % nb].ptr(), n * 4, h, 8), 2 * n * 4)
            for hints, tag in ((0, "plain"), (1, "nt_loads")):
                probe["read_%s_GBps" % tag] = rate(lambda r, h=hints: lib.adlhip_probe_read_ex(
                    d._h, bufs[r % nb].ptr(), n * 4, sink.ptr(), h, 8), n * 4)
            probe["copy_GBps"] = max(v for k, v in probe.items() if k.startswith("copy_"))
            probe["read_GBps"] = max(v for k, v in probe.items() if k.startswith("read_"))
            probe["buffers"] = "cold: %d key buffers of %d MiB in turns" % (nb, n * 4 >> 20)
        sink.release()

        # The literal "HBM-READ roofline" of the metric: the digit histogram (radix_count_kernel = the reference's StreamCountKernel,
        # RadixSort32Kernels.cl:176-236), a pure read stream of n keys -- measured where the library runs it on cold input: as the
        # first of the three launches of the multi-GPU partition (adlhip_partition_top_byte_u32), profiled per launch.
        hbm_read = None
        try:
            if nb >= 4:
                import ctypes as ct
                tb, wb = ct.c_size_t(), ct.c_size_t()
                lib.adlhip_radix_sort_scratch_bytes(d._h, 0, n, ct.byref(tb), ct.byref(wb))
                pw = Buffer(d, wb.value + wb.value // 8, np.uint8)
                pt = Buffer(d, 256, np.uint32)
                d.setParam("partition.lookback", 0)   # the three-kernel form: its first kernel is the plain digit histogram
                d.toggleProfiling(True)
                d.profile(reset=True)
                for r in range(8):
                    rc = lib.adlhip_partition_top_byte_u32(d._h, bufs[(2 * r) % nb].ptr(), bufs[(2 * r + 1) % nb].ptr(), pt.ptr(), pw.ptr(),
                                                           pw.getSize(), n)
                    if rc:
                        raise RuntimeError("adlhip_partition_top_byte_u32 failed")
                pprof = d.profile(reset=True)
                d.toggleProfiling(False)
                d.setParam("partition.lookback", 1)
                pw.release()
                pt.release()
                ck = [k for k in pprof if k.startswith("count_")]
                if ck:
                    ln, ms = pprof[ck[0]][0], pprof[ck[0]][1]
                    ach = n * ELEM_BYTES / (ms / ln * 1e-3) / 1e9
                    hbm_read = {"kernel": ck[0], "rocprof_kernel": "radix_count_kernel<unsigned int, 8, 256>", "launches": ln,
                                "avg_launch_ms": ms / ln, "algorithmic_bytes_per_launch": n * ELEM_BYTES, "achieved": ach,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "frac_of_read_probe": ach / probe["read_GBps"] if probe.get("read_GBps") else None,
                                "where": "first launch of adlhip_partition_top_byte_u32 on keys that are cold in every cache"}
        except Exception as e:
            hbm_read = {"error": repr(e)[:300]}

        algo = d.getParam("sort.algo")
        digit_bits = d.getParam("sort.digit_bits")
        for b in bufs:
            b.release()

        # BASELINE configs #3 and #5 beside the metric (not part of it): a few sorts each, fresh random inputs, device time
        # between two events on the library's stream; sortedness + checksum of one result checked outside the timing
        others = {}
        if n == N_KEYS and not args.no_other_configs:
            for label, nn, dtype, gen, unit, sorter in (
                    ("config3_key_value_64Mi_pairs", 1 << 26, np.uint64, 1, "Gpairs/s", lambda b, m: p.radixSort(d, b, m)),
                    ("config5_u64_256Mi_keys", 1 << 28, np.uint64, 2, "Gkeys/s", lambda b, m: p.radixSort64(d, b, m))):
                try:
                    reps = 4
                    bb = [Buffer(d, nn, dtype) for _ in range(reps + 1)]
                    for i, b in enumerate(bb):
                        b.generate(nn, seed=900 + i, kind=gen)
                    sorter(bb[0], nn)
                    DeviceUtils.waitForCompletion(d)
                    s3 = Stopwatch(d)
                    s3.start()
                    for b in bb[1:]:
                        sorter(b, nn)
                    s3.stop()
                    ms = s3.getMs() / reps
                    entry = {"value": nn / ms / 1e6, "unit": unit, "ms_per_sort": ms, "elements": nn, "sorts_timed": reps}
                    # every sweep of this sort priced as config #2's are: algorithmic bytes (8 read + 8 written per 8-byte element
                    # and sweep) over the launch's mean time between two events
                    for i, b in enumerate(bb[3:5]):   # (bb[1] is checked against the oracle below)
                        b.generate(nn, seed=950 + i, kind=gen)
                    d.toggleProfiling(True)
                    d.profile(reset=True)
                    for b in bb[3:5]:
                        sorter(b, nn)
                    oprof = d.profile(reset=True)
                    d.toggleProfiling(False)
                    rows = []
                    for k, v in oprof.items():
                        if k.startswith(("msd2s_pass", "msd2_pass", "msd2h_pass", "segment_sort", "onesweep_", "scatter_")) and v[1] / v[0] > 0.02:
                            by = 2 * nn * 8
                            ach = by / (v[1] / v[0] * 1e-3) / 1e9
                            row = {"kernel": k, "rocprof_kernel": PMC_KERNEL.get(k), "launches": v[0], "avg_launch_ms": v[1] / v[0],
                                   "algorithmic_bytes_per_launch": by, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
                            t = pmc_traffic.get(k) if isinstance(pmc_traffic, dict) else None
                            if isinstance(t, dict) and "traffic_bytes_per_launch" in t:
                                row["traffic"] = t["traffic_bytes_per_launch"]
                            rows.append(row)
                    rows.sort(key=lambda r: -r["avg_launch_ms"] * r["launches"])
                    if rows:
                        entry["roofline"] = dict(rows[0], bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", by_kernel=rows)
                    entry["kernels"] = {k: {"launches": v[0], "avg_ms": v[1] / v[0]} for k, v in oprof.items()}
                    if not args.no_verify:   # outside the timing: one result against the oracle, bit for bit
                        import oracle
                        got = bb[1].toHost()
                        want = (oracle.sort_kv32(oracle.pairs_kv32(nn, seed=901)) if gen == 1
                                else oracle.sort_u64(oracle.keys_u64(nn, seed=901)))
                        entry["verified_vs_oracle"] = bool(np.array_equal(got, want))
                        del got, want
                        if not entry["verified_vs_oracle"]:
                            raise SystemExit("bench: %s differs from the oracle" % label)
                    for b in bb:
                        b.release()
                    others[label] = entry
                except Exception as e:   # never lose the metric over an extra
                    others[label] = {"error": repr(e)[:300]}
        # config #3 once more with "sort.rank" = 0: ballot ranking everywhere, no reliance on the lane order of colliding returning DS
        # atomics (the hardware behaviour the default ranking rests on) -- what that guarantee costs
        if n == N_KEYS and not args.no_other_configs and "error" not in others.get("config3_key_value_64Mi_pairs", {"error": 1}):
            try:
                nn = 1 << 26
                d.setParam("sort.rank", 0)
                bb = [Buffer(d, nn, np.uint64) for _ in range(4)]
                for i, b in enumerate(bb):
                    b.generate(nn, seed=970 + i, kind=1)
                p.radixSort(d, bb[0], nn)
                DeviceUtils.waitForCompletion(d)
                s4 = Stopwatch(d)
                s4.start()
                for b in bb[1:]:
                    p.radixSort(d, b, nn)
                s4.stop()
                ms = s4.getMs() / 3
                got = bb[1].toHost()
                kk = got & np.uint64(0xffffffff)
                vv = got >> np.uint64(32)
                same = kk[1:] == kk[:-1]
                ok = bool(np.all(kk[1:] >= kk[:-1])) and bool(np.all(vv[1:][same] > vv[:-1][same]))
                others["config3_rank0_ballot_ranking"] = {"value": nn / ms / 1e6, "unit": "Gpairs/s", "ms_per_sort": ms, "elements": nn,
                                                          "sorted_and_stable": ok}
                del got, kk, vv, same
                for b in bb:
                    b.release()
            except Exception as e:
                others["config3_rank0_ballot_ranking"] = {"error": repr(e)[:300]}
            finally:
                d.setParam("sort.rank", d.getParam("sort.lds_ordered"))
        # SURVEY f3, second half: 64 Mi u32 keys with u64 VALUES on separate arrays (index sort + one gather, adlhip_radix_sort_soa)
        if n == N_KEYS and not args.no_other_configs:
            try:
                nn = 1 << 26
                kb = Buffer(d, nn, np.uint32)
                vb = Buffer(d, nn, np.uint64)
                ts = []
                for t in range(3):
                    kb.generate(nn, seed=980 + t)
                    vb.generate(nn, seed=990 + t, kind=2)
                    DeviceUtils.waitForCompletion(d)
                    s5 = Stopwatch(d)
                    s5.start()
                    p.radixSortSoA(d, kb, vb, nn)
                    s5.stop()
                    ts.append(s5.getMs())
                got = kb.toHost()
                others["f3_u32_keys_u64_values_64Mi"] = {"value": nn / min(ts[1:]) / 1e6, "unit": "Gpairs/s", "ms_per_sort": min(ts[1:]), "elements": nn,
                                                         "keys_sorted": bool(np.all(got[1:] >= got[:-1]))}
                del got
                kb.release()
                vb.release()
            except Exception as e:
                others["f3_u32_keys_u64_values_64Mi"] = {"error": repr(e)[:300]}
        out["other_configs"] = others
        # The same sort on keys that are NOT uniform (beside the metric, not part of it): the large sort's slabs give every bucket the
        # same room, so such keys end in its safety net (counting sort for few distinct values, LSD passes otherwise) -- the bench must
        # not only show the best case.  Each row: the first sort of a FRESH handle and its third, ms per sort, result checked.
        dists = {}
        if n == N_KEYS and not args.no_other_configs:
            try:
                ii = np.arange(n, dtype=np.uint32)
                hh = ii * np.uint32(2654435761)
                kinds = [("sorted", None), ("all_equal", np.full(n, 0x12345678, dtype=np.uint32)),
                         ("256_values", ((hh >> np.uint32(24)) * np.uint32(0x01010101)) ^ np.uint32(0x5a5a0000)),
                         ("4096_values", (hh >> np.uint32(20)) * np.uint32(0x00100801)),
                         ("heavy_top_byte", np.where(ii % 10 != 0, (hh >> np.uint32(8)) | np.uint32(0x37000000), hh * np.uint32(40503)).astype(np.uint32))]
                for name, host in kinds:
                    d2 = DeviceUtils.allocate()
                    p2 = Pprims()
                    p2.reserve(d2, 0, n)
                    b2 = Buffer(d2, n, np.uint32)
                    if host is None:
                        b2.generate(n, seed=77)
                        p2.radixSort(d2, b2, n)
                        DeviceUtils.waitForCompletion(d2)
                        host = b2.toHost()
                    ts = []
                    for t in range(3):
                        b2.write(host)
                        DeviceUtils.waitForCompletion(d2)
                        sw2 = Stopwatch(d2)
                        sw2.start()
                        p2.radixSort(d2, b2, n)
                        sw2.stop()
                        ts.append(sw2.getMs())
                    got = b2.toHost()
                    ok = bool(np.all(got[1:] >= got[:-1])) and int(got.astype(np.uint64).sum()) == int(host.astype(np.uint64).sum())
                    dists[name] = {"first_ms": ts[0], "third_ms": ts[2], "sorted_and_sum_preserved": ok,
                                   "net_runs": d2.getParam("stat.net_runs"), "net_counting": d2.getParam("stat.net_counting")}
                    b2.release()
                    p2.close()
                    DeviceUtils.deallocate(d2)
                    del got
                    if not ok:
                        raise SystemExit("bench: wrong result on %s keys" % name)
            except SystemExit:
                raise
            except Exception as e:
                dists["error"] = repr(e)[:300]
        out["distributions_64Mi_u32"] = dists
        # ... and {key, value} pairs (config #3's shape) whose keys repeat: group-by keys with a payload.  At most 256 values: one
        # stable pass on the key's rank in a dictionary, inside the net; more: its four LSD passes.  Values = input positions, so
        # stability is checked exactly: ascending inside every run of equal keys.
        pdists = {}
        if n == N_KEYS and not args.no_other_configs:
            try:
                ii = np.arange(n, dtype=np.uint32)
                hh = ii * np.uint32(2654435761)
                for name, keys in (("16_values", (hh >> np.uint32(28)) * np.uint32(0x11111111)),
                                   ("256_values", ((hh >> np.uint32(24)) * np.uint32(0x01010101)) ^ np.uint32(0x5a5a0000)),
                                   ("4096_values", (hh >> np.uint32(20)) * np.uint32(0x00100801))):
                    host = keys.astype(np.uint64) | (ii.astype(np.uint64) << np.uint64(32))
                    d2 = DeviceUtils.allocate()
                    p2 = Pprims()
                    p2.reserve(d2, 1, n)
                    b2 = Buffer(d2, n, np.uint64)
                    ts = []
                    for t in range(3):
                        b2.write(host)
                        DeviceUtils.waitForCompletion(d2)
                        sw2 = Stopwatch(d2)
                        sw2.start()
                        p2.radixSort(d2, b2, n)
                        sw2.stop()
                        ts.append(sw2.getMs())
                    got = b2.toHost()
                    gk = (got & np.uint64(0xffffffff)).astype(np.uint32)
                    gv = (got >> np.uint64(32)).astype(np.uint32)
                    ok = bool(np.all(gk[1:] >= gk[:-1])) and bool(np.all((gv[1:] > gv[:-1]) | (gk[1:] != gk[:-1]))) and bool(np.array_equal(keys[gv], gk))
                    pdists[name] = {"first_ms": ts[0], "third_ms": ts[2], "sorted_stable_and_a_permutation": ok,
                                    "net_runs": d2.getParam("stat.net_runs"), "net_counting": d2.getParam("stat.net_counting")}
                    b2.release()
                    p2.close()
                    DeviceUtils.deallocate(d2)
                    del got, gk, gv, host
                    if not ok:
                        raise SystemExit("bench: wrong result on pairs with %s" % name)
            except SystemExit:
                raise
            except Exception as e:
                pdists["error"] = repr(e)[:300]
        out["distributions_64Mi_pairs"] = pdists
        p.close()
        info_name = d.getDeviceName()
        DeviceUtils.deallocate(d)

        # Every sweep of the sort priced on its own (pass 1 / pass 2 / finish of the large sort; the per-digit passes otherwise)
        # by the ALGORITHMIC bytes of one launch (what the step must read and write): a pass reads and
        # writes the array; in the large sort the second slab holds the keys' low 16 bits only, so its second pass writes 2 bytes
        # per key and its finish reads 2.
        large = "msd2_pass1_u32" in prof
        def alg_bytes(k):
            if k == "msd2_pass2_u32":
                return n * (ELEM_BYTES + 2)
            if k.startswith("segment_sort") and large:
                return n * (2 + ELEM_BYTES)
            return 2 * n * ELEM_BYTES
        groups = {}   # launch name -> (launches, ms, algorithmic bytes over all launches)
        for k, v in prof.items():
            if k.startswith(("onesweep_", "scatter_", "segment_sort", "msd2_pass", "msd2s_pass")):
                groups[k] = (v[0], v[1], v[0] * alg_bytes(k))
        dom_name, (dom_launches, dom_ms, dom_bytes) = max(groups.items(), key=lambda kv: kv[1][1])
        dom_avg_s = dom_ms / dom_launches * 1e-3
        achieved = dom_bytes / dom_launches / dom_avg_s / 1e9
        # HBM traffic per launch from the PMC counters, measured now: two child runs of this file under rocprofv3 (one counter
        # each, kernel trace only -- MI355X_MICROARCH.md's recipe); the committed figures of the round's profiling session
        # (profiles/*pmc_traffic.json) stand in only when rocprofv3 cannot be run here
        traffic, traffic_src, traffic_all = None, None, pmc_traffic
        if isinstance(traffic_all.get(dom_name), dict) and "traffic_bytes_per_launch" in traffic_all[dom_name]:
            traffic = traffic_all[dom_name]["traffic_bytes_per_launch"]
            traffic_src = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, two child runs of this file "
                           "started before the timed part")
        if traffic is None:
            try:
                import glob
                for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
                    t = json.load(open(f))
                    if (t.get("profile_name") == dom_name and t.get("keys_per_launch") == n
                            and abs(t.get("algorithmic_bytes_per_launch", 0) - dom_bytes / dom_launches) < 1e-3 * dom_bytes / dom_launches):
                        traffic, traffic_src = t["traffic_bytes_per_launch"], os.path.relpath(f, ROOT) + " (committed; not measured in this run)"
                        break
            except Exception:
                pass
        by_kernel = []
        for k, (ln, ms, by) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            ach = by / ln / (ms / ln * 1e-3) / 1e9
            row = {"kernel": k, "launches": ln, "avg_launch_ms": ms / ln, "algorithmic_bytes_per_launch": by / ln,
                   "achieved": ach, "frac": ach / HBM_PEAK_GBS}
            if isinstance(traffic_all.get(k), dict) and "traffic_bytes_per_launch" in traffic_all[k]:
                row["traffic"] = traffic_all[k]["traffic_bytes_per_launch"]
                row["fetch_kb"] = traffic_all[k]["FETCH_SIZE_kb"]
                row["write_kb"] = traffic_all[k]["WRITE_SIZE_kb"]
            by_kernel.append(row)
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": dom_name, "rocprof_kernel": PMC_KERNEL.get(dom_name), "launches": dom_launches, "avg_launch_ms": dom_ms / dom_launches,
            "algorithmic_bytes_per_launch": dom_bytes / dom_launches,
            "timing": "hipEvent pair around every launch on the library's stream, second loop over the same K inputs",
            "by_kernel": by_kernel,
        }
        if "error" in traffic_all:
            out["roofline"]["traffic_error"] = traffic_all["error"]
        if probe.get("copy_GBps"):
            out["roofline"]["frac_of_copy"] = achieved / probe["copy_GBps"]
            for row in by_kernel:
                row["frac_of_copy"] = row["achieved"] / probe["copy_GBps"]
        out["roofline"]["hbm_read"] = hbm_read
        out["kernels"] = {k: {"launches": v[0], "avg_ms": v[1] / v[0]} for k, v in prof.items()}
        out["probe"] = probe
        out["device"] = info_name
        out["verified_vs_oracle"] = verified
        out["event_ms_per_step"] = ev_ms / K
        # whole sort on the bytes it really moves: one histogram read (if the path has one) + (read + write) per global pass
        passes = max(1, round(sum(v[0] for v in groups.values()) / K))
        hist_reads = 1 if any(k.startswith(("os_hist_", "count_", "mid_prep")) for k in prof) else 0
        moved = ELEM_BYTES * hist_reads + sum(v[2] for v in groups.values()) / K / n
        out["whole_sort"] = {"global_passes": passes, "histogram_reads": hist_reads, "bytes_moved_per_key": moved,
                             "achieved_GBps": n * moved / (ev_ms / K * 1e-3) / 1e9,
                             "frac_of_peak": n * moved / (ev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS}
        parallelism = "1 GPU"
        workload = "Key-only RadixSort32, %d uniform-random u32 keys (splitmix64 hi32, seed 123+step), 1xMI355X, in place" % n
        auto = "auto (two MSD bucket passes + LDS finish at this size)" if large else "auto (one-sweep at this size)"
        algo_name = {0: "onesweep", 1: "three-kernel", -1: auto}.get(algo, str(algo))
        cfg_extra = {"sort_algo": algo_name, "digit_bits": digit_bits,
                     "clock_settle": {"untimed_sorts_before_warmup": settle_sorts,
                                      "why": "the chip needs ~10 ms of sustained load after an idle gap to reach its steady clock; "
                                             "the LDS-bound finish runs 20 % slower until then (profiles/r4_clock_ramp.txt)"}}
    else:
        from oclradixsort_amd.dist import HipBackend, ShardedRadixSort
        be = HipBackend(local_rank)
        if args.algo is not None:
            be.setParam("sort.algo", args.algo)
        if args.digit_bits is not None:
            be.setParam("sort.digit_bits", args.digit_bits)
        # ADLHIP_BENCH_FORCE_BUCKETS=8 with ADLHIP_BENCH_FORCE_DIST=1: pay an 8-rank partition pass on the 1-GPU rehearsal
        sorter = ShardedRadixSort(be, rehearse_buckets=int(os.environ.get("ADLHIP_BENCH_FORCE_BUCKETS", "0")) if force_dist else 0)

        def make_inputs(nk, count, seed0):
            ts = []
            for i in range(count):
                t = be.empty(nk)
                b = Buffer(dtype=np.uint32)
                b.setRawPtr(be.device, t.data_ptr(), nk)
                # global index space: rank r owns indices [r*nk, (r+1)*nk) of step i's key sequence
                b.generate(nk, seed=seed0 + i, firstIndex=rank * nk)
                ts.append(t)
            return ts

        def run(batches):
            # the independent batches go through the two-stage pipeline (dist.py: batch i+1's partition +
            # all-to-all overlap batch i's local sort); fill and drain of the pipeline are inside the timed region
            last = None
            for last in sorter.sort_stream(batches, force_exchange=force_dist):
                pass
            return last

        def verify(res, nk, seed_last):
            """Size-independent checks on the last batch: local sortedness, ownership (this rank's top bytes lie inside
            its splitter range, or equal its fixed bucket), global count and checksum conservation."""
            x = res.view(torch.int32) ^ (-2147483648)           # order-preserving u32 -> i32 map
            ok = bool((x[1:] >= x[:-1]).all().item()) if x.numel() > 1 else True
            top = (res.to(torch.int64) & 0xffffffff) >> 24
            if sorter.last_bounds is not None and res.numel():
                # sort_stream has already partitioned nothing newer than the last batch: these are its bounds
                b = sorter.last_bounds
                ok = ok and bool(((top >= b[rank]) & (top < b[rank + 1])).all().item())
            elif res.numel():
                ok = ok and bool(((top >> (8 - (world.bit_length() - 1))) == rank).all().item())
            cnt = torch.tensor([res.numel(), int((res.to(torch.int64) & 0xffffffff).sum().item())],
                               dtype=torch.int64, device=res.device)
            dist.all_reduce(cnt)
            ref_t = make_inputs(nk, 1, seed_last)[0]
            torch.cuda.synchronize()
            ref_cnt = torch.tensor([nk, int((ref_t.to(torch.int64) & 0xffffffff).sum().item())],
                                   dtype=torch.int64, device=res.device)
            dist.all_reduce(ref_cnt)
            ok = ok and bool((cnt == ref_cnt).all().item())
            flag = torch.tensor([1 if ok else 0], device=res.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        inputs = make_inputs(n, K + W, 123)
        be.reserve(n)   # slots and scratch of the pipeline: allocated before anything is timed, whatever W is
        torch.cuda.synchronize()
        res = run(inputs[:W])
        barrier()
        t0 = time.perf_counter()
        res = run(inputs[W:W + K])
        barrier()
        wall = time.perf_counter() - t0
        share = torch.tensor([res.numel()], dtype=torch.int64, device=res.device)
        if dist.is_initialized():
            dist.all_reduce(share, op=dist.ReduceOp.MAX)
        out["max_rank_share_over_mean"] = float(share.item()) / float(n)

        # for the record, not the metric: the same K batches one at a time (partition -> all-gather -> all-to-all ->
        # local sort back to back on one stream), with a per-stage breakdown from events on this rank
        marks = []
        for i in range(min(W, 2)):
            sorter.sort(inputs[i], force_exchange=force_dist)
        barrier()
        t1 = time.perf_counter()
        for i in range(W, W + K):
            sorter.sort(inputs[i], force_exchange=force_dist, marks=marks)
        barrier()
        serial_wall = time.perf_counter() - t1
        st = [0.0, 0.0, 0.0, 0.0]
        for i in range(0, len(marks), 5):
            for j in range(4):
                st[j] += marks[i + j].elapsed_time(marks[i + j + 1])
        out["serial"] = {
            "ms_per_step": serial_wall / K * 1e3,
            "Gkeys_per_s": float(n) * world * K / serial_wall / 1e9,
            "rank0_stage_ms": {"partition_and_splitters": st[0] / K, "count_allgather_and_host_sync": st[1] / K,
                               "all_to_all": st[2] / K, "local_sort": st[3] / K},
            # bytes this rank put on the links per step (everything but its own bucket) / the all-to-all's duration
            "rank0_all_to_all_out_GBps": (sum(sorter.last_splits[0]) - sorter.last_splits[0][rank]) * 4.0 / max(st[2] / K * 1e-3, 1e-12) / 1e9,
            "note": "un-pipelined driver (ShardedRadixSort.sort), timed after the metric's region",
        }
        verified = None
        if not args.no_verify:
            res = run(inputs[W + K - 1:W + K])      # bounds and result of the same (last) batch
            verified = verify(res, n, 123 + W + K - 1)
            if not verified:
                raise SystemExit("bench: multi-GPU result failed sortedness/ownership/checksum checks")
        out["verified_properties"] = verified
        del inputs, res

        # BASELINE config #4 at its stated size, beside the weak-scaling metric: 2^30 keys in total over the ranks
        # (2^27 per GPU at N = 8).  A few steps through the same pipeline; not the line's `value`.
        if os.environ.get("ADLHIP_BENCH_CONFIG4", "1") != "0" and (world > 1 or os.environ.get("ADLHIP_BENCH_CONFIG4") == "1"):
            try:
                n4 = int(os.environ.get("ADLHIP_BENCH_CONFIG4_TOTAL", CONFIG4_TOTAL_KEYS)) // world
                K4, W4 = 3, 1
                torch.cuda.empty_cache()
                in4 = make_inputs(n4, K4 + W4, 900)
                be.reserve(n4)
                torch.cuda.synchronize()
                r4 = run(in4[:W4])
                barrier()
                t4 = time.perf_counter()
                r4 = run(in4[W4:])
                barrier()
                w4 = time.perf_counter() - t4
                if dist.is_initialized():
                    tt = torch.tensor([w4], dtype=torch.float64, device="cuda")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    w4 = float(tt.item())
                v4 = None
                if not args.no_verify:
                    r4 = run(in4[-1:])
                    v4 = verify(r4, n4, 900 + K4 + W4 - 1)
                    if not v4:
                        raise RuntimeError("config #4 result failed sortedness/ownership/checksum checks")
                out["config4"] = {
                    "workload": "BASELINE config #4: %d u32 keys sharded across %d x MI355X (%d per GPU), top-byte partition + balanced "
                                "splitters + RCCL all-to-all + local RadixSort32" % (n4 * world, world, n4),
                    "keys_per_gpu": n4, "total_keys": n4 * world, "steps": K4, "warmup": W4,
                    "ms_per_step": w4 / K4 * 1e3, "Gkeys_per_s": float(n4) * world * K4 / w4 / 1e9,
                    "scaling": "strong (total fixed at 2^30 keys)", "verified_properties": v4,
                }
                del in4, r4
            except Exception as e:   # never lose the metric line to the side measurement
                out["config4"] = {"error": repr(e)[:300]}
        be.close()
        parallelism = ("top-byte buckets with balanced splitters x%d (all-to-all over RCCL), exchange of batch i+1 overlapped "
                       "with local sort of batch i" % world)
        workload = ("%d uniform-random u32 keys per GPU (weak scaling), top-byte partition + balanced splitters + all-to-all + "
                    "local RadixSort32" % n)
        cfg_extra = {}

    # max over ranks
    if dist.is_initialized():
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    if rank == 0:
        total_keys = float(n) * world * K
        value = total_keys / wall / 1e9
        line = {
            "metric": "Gkeys/s on uniform-random u32 keys; achieved % of HBM-read roofline",
            "value": value,
            "unit": "Gkeys/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": wall / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": dict({"workload": workload, "keys_per_gpu": n, "parallelism": parallelism}, **cfg_extra),
        }
        line.update(out)
        if rehearse:
            line["data"] = "REHEARSAL on one GPU with host-staged collectives: not a benchmark number"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], _ = cpu_baseline(n)
    else:
        line = None

    if dist.is_initialized():
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints a version banner through C stdio, which is flushed only at exit: push it out first so
        # that the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
