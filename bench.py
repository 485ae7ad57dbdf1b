#!/usr/bin/env python3
"""bench.py -- throughput of the receive-DSP hot path on MI355X.

Headline workload (BASELINE.json configs[1]): 64 SSB receivers per GPU on one 1.536 Msps complex-double I/Q stream,
cascaded decimating FIRs, SSB demodulation, AGC, fp64.  A "step" is one pass of the whole receive chain
(quisk_process_samples for every receiver) over one block of `--block` synthetic I/Q samples that are already
resident in HBM.  value = input samples processed per second (summed over ranks) / 1e6.

Further legs in the same JSON line (each with its own roofline object):
  config4_slice  the per-GPU share of configs[3]: 128 mixed SSB/AM/FM receivers at 61.44 Msps, 2^20-sample blocks; with
                 N > 1 ranks the audio of all receivers is gathered to rank 0 over RCCL and that gather is timed
  config5_slice  the per-GPU share of configs[4]: the same slice in fp32 mode with a 2^20-point FFT of every block beside it
  spectrum       configs[2]: 2^20-point windowed, averaged display FFT at 61.44 Msps (rank 0), CPU FFT timed beside it
  config4_one_gpu  the WHOLE of configs[3] on one GPU: 1024 mixed receivers at 61.44 Msps with the fp64 2^20-point FFT of every
                 block beside them -- does the north-star target fit one MI355X in real time? (N = 1 only)
  config1        configs[0]: 1 USB receiver at 192 ksps in 960-sample blocks, engine beside the single-threaded CPU port
  cpu_baseline   the CPU port (oracle) of the headline workload on the host cores (rank 0, N = 1 only)

Multi-GPU: `python bench.py --gpus N` with no RANK in the environment starts N ranks itself (torch.distributed.run,
one process per GPU, before anything here touches a GPU); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.
Receivers are sharded, `--receivers` per rank, every rank reads the same I/Q stream (weak scaling, no data-path
collective).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FE_FLOPS_PER_SAMPLE = {0: 12.0, 1: 12 + 34.0, 2: 12 + 34 * 1.5, 3: 12 + 34 * 1.75, 4: 12 + 34 * 1.875, 5: 12 + 34 * 1.9375}
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak: half the 157.3 TF FP32 vector rate of MI355X_MICROARCH.md
FP32_VALU_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 200 blocks of 2^20 samples = 0.08 s of GPU time: fill and drain of the
    ap.add_argument("--warmup", type=int, default=10)     # three-deep pipeline are 1 % of it (they are 4 % of 40 steps)
    ap.add_argument("--block", type=int, default=1 << 20, help="I/Q samples per step")
    ap.add_argument("--receivers", type=int, default=64, help="receivers per GPU")
    ap.add_argument("--rate", type=int, default=1536000)
    ap.add_argument("--mix", default="ssb", choices=["ssb", "mixed"])
    ap.add_argument("--fp32", action="store_true", help="run the headline leg in the engine's fp32 mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spectrum", action="store_true", help="skip the configs[2] spectrum leg")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[3] per-GPU slice leg")
    ap.add_argument("--no-config1", action="store_true", help="skip the configs[0] leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the configs[4] fp32 slice leg")
    ap.add_argument("--no-one-gpu", action="store_true", help="skip the 1024-receivers-on-one-GPU leg")
    ap.add_argument("--cpu-samples", type=int, default=1 << 21)
    ap.add_argument("--preroll-ms", type=float, default=30.0, help="run the leg's blocks untimed for this long before the W warm-up steps (clock conditioning)")
    return ap.parse_args()


def launch_ranks(a):
    """`--gpus N` without a launcher: become the launcher.  Nothing in this process has touched a GPU yet."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


# ------------------------------------------------------------------------------------------------ CPU legs
def cpu_baseline(fs, modes, tunes, n_samples, seed, note=""):
    """The oracle (CPU restatement of the reference's per-sample loops, kind "port") on the host cores:
    one receiver per thread, blocks of 65536 like config 2.  Test infrastructure used as the checker's clock only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    from _rxsetup import make_oracle_rx
    from quisk_amd.synth import make_iq
    cores = max(1, min(16, os.cpu_count() or 1, len(modes)))
    x = make_iq(n_samples, fs, tunes[:8], modes[:8], seed)
    rxs = [make_oracle_rx(fs, int(m), int(t)) for m, t in zip(modes, tunes)]

    def run(o):
        for i in range(0, n_samples, 65536):
            o.process(x[i:i + 65536])

    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(run, rxs))
    dt = time.time() - t0
    for o in rxs:
        o.close()
    return {"value": n_samples / dt / 1e6, "unit": "input Msamples/s at %d receivers" % len(modes), "cores": cores,
            "kind": "port", "realtime_factor": n_samples / dt / fs,
            "sample": "%d receivers x %d I/Q samples (%.4f s of signal) in 65536-sample blocks, one receiver per thread%s"
                      % (len(modes), n_samples, n_samples / fs, note)}


def reference_calibration(n=1 << 21):
    """What "kind: port" is worth against the reference's own code: the decimation chain of config 2 -- quisk_cDecim2HB45 x 4
    then quisk_cDecimate with the 98-tap filter, / 2 (filter.c:354-394, 180-206; quisk.c:1649-1668, 1706) -- over the same
    samples on one thread, once through oracle/_ref (the reference's filter.c compiled unmodified; it travels with the tree as
    a built library) and once through the port.  None where oracle/_ref is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import _bind as B
    if not B.have_ref():
        return None
    rng = np.random.default_rng(7)
    x = np.round((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 2.0 ** 20)
    h = B.oracle_table("qk_h_dec2_98")
    R, O = B.ref(), B.oracle()

    def run_ref():
        w = x.copy(); m = n
        hbs = [B.RCHB() for _ in range(4)]
        f = B.RCFilter()
        R.quisk_filt_cInit(C.byref(f), h.ctypes.data_as(B.c_dp), len(h))
        t0 = time.perf_counter()
        for hb in hbs:
            m = R.quisk_cDecim2HB45(w, m, C.byref(hb))
        m = R.quisk_cDecimate(w, m, C.byref(f), 2)
        return time.perf_counter() - t0, w[:m].copy()

    def run_port():
        w = x.copy(); m = n
        hbs = [B.OCHB() for _ in range(4)]
        f = B.OCFir()
        O.qko_cfir_init(C.byref(f), h.ctypes.data_as(B.c_dp), len(h))
        t0 = time.perf_counter()
        for hb in hbs:
            m = O.qko_cdecim2hb45(w, m, C.byref(hb))
        m = O.qko_cdecimate(w, m, C.byref(f), 2)
        return time.perf_counter() - t0, w[:m].copy()

    run_ref(); run_port()
    tr, yr = min((run_ref() for _ in range(3)), key=lambda p: p[0])
    tp, yp = min((run_port() for _ in range(3)), key=lambda p: p[0])
    return {"chain": "quisk_cDecim2HB45 x 4 -> quisk_cDecimate(98 taps, / 2), %d complex samples, one thread, best of 3" % n,
            "reference_msamples_per_s": n / tr / 1e6, "port_msamples_per_s": n / tp / 1e6, "port_over_reference": tr / tp,
            "kind": "reference", "bit_identical": bool(np.array_equal(yr.view(np.float64), yp.view(np.float64)))}


def rocprof_kernel_ms(summary_csv, needle):
    """Average duration (ms) of the kernel whose name contains `needle` in a committed rocprofv3 --kernel-trace --stats summary
    (profiles/*.csv), or None."""
    import csv
    path = os.path.join(ROOT, "profiles", summary_csv)
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if needle in r.get("Name", ""):
            return float(r["AverageNs"]) / 1e6
    return None


def cpu_fft_baseline(N, frames_single=3, frames_multi=16):
    """The oracle's FFT (FFTW is not in the image; the reference's FFT cannot travel) with the Hann window and the
    magnitude accumulation of get_graph: frames/s on one thread and on all the box's cores (one frame per thread)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import _bind as B
    rng = np.random.default_rng(1003)
    x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) * 2.0 ** 20
    cores = max(1, min(16, os.cpu_count() or 1))

    def frame(_):
        g = B.OracleGraph(N, 4096, 61440000)
        g.add_frames(x)
        g.close()

    frame(0)
    t0 = time.time()
    for i in range(frames_single):
        frame(i)
    single = frames_single / (time.time() - t0)
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(frame, range(frames_multi)))
    multi = frames_multi / (time.time() - t0)
    return {"value": multi, "unit": "frames/s", "cores": cores, "kind": "port", "single_thread_frames_per_s": single,
            "sample": "%d + %d frames of %d points: window, FFT (the oracle's radix-2; FFTW is absent), |X| accumulation"
                      % (frames_single, frames_multi, N)}


# ------------------------------------------------------------------------------------------------ GPU legs
def make_engine(fs, nrx, block, local, tunes, modes, fp32=False, **kw):
    from quisk_amd import Engine, rx_filter
    eng = Engine(fs, n_receivers=nrx, max_block=block, device=local, fp32=fp32, **kw)
    for r in range(nrx):
        fi, fq, bw, off = rx_filter(fs, int(modes[r]))
        eng.set_mode(r, int(modes[r]))
        eng.set_tune(r, int(tunes[r]))
        eng.set_filters(r, fi, fq, bw, off)
    return eng


PREROLL_MS = 30.0


def timed_leg(eng, ptrs, block, steps, warmup, barrier, timing_groups, preroll_ms=None):
    """W untimed steps, then exactly K steps between barriers; returns (seconds, stats)."""
    nb = len(ptrs)
    if preroll_ms is None:
        preroll_ms = PREROLL_MS
    if preroll_ms > 0:
        # clock conditioning, not part of the measurement: the same blocks for PREROLL_MS of wall time, so that the W + K steps
        # below run at the clocks of a stream that has been running (the GPU's clocks take ~20 ms of load to come up from idle)
        t_end = time.perf_counter() + preroll_ms * 1e-3
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(8):
                eng.process_device(ptrs[i % nb], block); i += 1
            eng.sync()
    for i in range(warmup):
        eng.process_device(ptrs[i % nb], block)
    eng.sync()
    eng.reset_stats()
    eng.set_timing(True, groups=timing_groups)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.process_device(ptrs[i % nb], block)
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    eng.set_timing(False)
    return dt, st


def frontend_roofline(st, fs, block, nrx, dt_step_ms, fp32, pmc_name, rocprof_csv=None):
    from quisk_amd import plan_decimation
    from quisk_amd.engine import KT_FRONTEND
    _, d2, _, _ = plan_decimation(fs)
    S = min(max(d2 - 1, 0), 5)
    fe_ms = st.kernel_ms_sum[KT_FRONTEND] / max(1, st.kernel_launches[KT_FRONTEND])
    fe_flops = FE_FLOPS_PER_SAMPLE[S] * block * nrx
    achieved_tf = fe_flops / (fe_ms * 1e-3) / 1e12 if fe_ms > 0 else 0.0
    peak = FP32_VALU_PEAK_TFLOPS if fp32 else FP64_VALU_PEAK_TFLOPS
    traffic, source = None, None
    pmc = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            source = "profiles/%s (rocprofv3 --pmc passes of an earlier run of this command, not this run)" % pmc_name
        except Exception:
            traffic = None
    esz = 8 if fp32 else 16
    # the same kernel's average duration in the committed rocprofv3 summary of this command: the HIP-event figure includes the
    # wait of the second event behind other queues' packets, the profiler's is the kernel's own begin-to-end time
    needle = ("k_frontend_pk<%d>" % S) if (fp32 and os.environ.get("QK_FE_PACKED", "1") != "0") else "k_frontend_wave<%d, %s>" % (S, "float" if fp32 else "double")
    rp_ms = rocprof_kernel_ms(rocprof_csv, needle) if rocprof_csv else None
    rp = None
    if rp_ms:
        rp = {"avg_launch_ms": rp_ms, "achieved": fe_flops / (rp_ms * 1e-3) / 1e12, "frac": fe_flops / (rp_ms * 1e-3) / 1e12 / peak,
              "source": "profiles/%s (rocprofv3 --kernel-trace --stats of an earlier run of this command)" % rocprof_csv}
    return {"kernel": ("k_frontend_pk<S=%d> (packed-FP32 mixer + %d fused half-band decimators)" if needle.startswith("k_frontend_pk") else
                       "k_frontend_wave<S=%d> (mixer + %d fused half-band decimators)") % (S, S), "rocprof": rp,
            "bound": "valu", "pipe": "fp32 vector FMA" if fp32 else "fp64 vector FMA", "achieved": achieved_tf, "peak": peak,
            "unit": "TFLOP/s", "frac": achieved_tf / peak, "traffic": traffic, "traffic_source": source,
            "avg_launch_ms": fe_ms, "launches": int(st.kernel_launches[KT_FRONTEND]),
            "algorithmic_flop_per_launch": fe_flops,
            "hbm": {"achieved": (16.0 * block + esz * (block >> S) * nrx) / (fe_ms * 1e-3) / 1e9 if fe_ms > 0 else 0.0,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            "share_of_step": fe_ms / dt_step_ms if dt_step_ms > 0 else None}


def spectrum_leg(local, dev_block_ptrs, block, with_cpu):
    """BASELINE configs[2]: 2^20-point periodic-Hann windowed FFT at 61.44 Msps, magnitude average of 8 frames
    released as a 4096-point dB row (get_graph).  Frames come from the blocks already resident in HBM.
    Reported: frames/s, the algorithmic HBM rate (16 B per sample read once) and the FFT kernels' own time."""
    import torch
    from quisk_amd import Engine
    from quisk_amd.engine import KT_FFT
    N = 1 << 20
    nb = len(dev_block_ptrs)
    if block != N or nb < 16:           # (the resident blocks are contiguous: eight of them are one 8-frame block)
        return None
    eng = Engine(61440000, n_receivers=0, max_block=8 * N, fft_size=N, data_width=4096, device=local)
    row = None
    eng.process_device(dev_block_ptrs[0], 8 * N)
    _, row, _ = eng.get_graph(1.0, 0.0)
    torch.cuda.synchronize()
    groups, t0 = 24, time.perf_counter()
    rows = 0
    for g in range(groups):             # blocks of eight frames (0.137 s of signal each), one averaged row released per block:
        eng.process_device(dev_block_ptrs[(g % 2) * 8], 8 * N)      # the row of block g is marked behind its frames and taken
        if g:                                                       # by the host while block g + 1 is under way
            cnt, row, _ = eng.get_graph(1.0, 0.0)
            rows += cnt == 8
        eng.graph_mark(1.0, 0.0)
    cnt, row, _ = eng.get_graph(1.0, 0.0)
    rows += cnt == 8
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rows != groups:
        raise SystemExit("spectrum leg: %d rows of 8 frames expected, %d released" % (groups, rows))
    t2 = time.perf_counter()
    for g in range(groups):             # the same with the row taken before the next block is queued (no mark)
        eng.process_device(dev_block_ptrs[(g % 2) * 8], 8 * N)
        eng.get_graph(1.0, 0.0)
    torch.cuda.synchronize()
    fps_sync = groups * 8 / (time.perf_counter() - t2)
    t1 = time.perf_counter()
    for g in range(6):                  # the same with one frame per block
        for i in range(8):
            eng.process_device(dev_block_ptrs[i], N)
        eng.get_graph(1.0, 0.0)
    torch.cuda.synchronize()
    fps_single = 48 / (time.perf_counter() - t1)
    # the kernels' own time, from HIP events on the stream they run on (a second, shorter pass: events cost queue time)
    eng.reset_stats()
    eng.set_timing(True, groups=(KT_FFT,))
    for g in range(4):
        eng.process_device(dev_block_ptrs[(g % 2) * 8], 8 * N)
    eng.get_graph(1.0, 0.0)
    st = eng.stats()
    eng.set_timing(False)
    k_ms = st.kernel_ms_sum[KT_FFT] / max(1, st.kernel_launches[KT_FFT]) / 8.0      # per frame: a launch pair takes eight
    eng.close()
    fps = groups * 8 / dt
    gbs = fps * N * 16.0 / 1e9
    k_gbs = N * 16.0 / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, source = None, None
    pmc = os.path.join(ROOT, "profiles", "r03_pmc_fft.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_frame")
            source = "profiles/r03_pmc_fft.json (rocprofv3 --pmc passes of an earlier run, not this run)"
        except Exception:
            traffic = None
    res = {"workload": "BASELINE configs[2]: 2^20-point windowed FFT at 61.44 Msps, 8-frame magnitude average, 4096-point dB row",
           "frames_per_s": fps, "frames_per_s_row_taken_before_next_block": fps_sync, "frames_per_s_one_frame_blocks": fps_single, "realtime_frames_per_s": 61440000.0 / N,
           "msamples_per_s": fps * N / 1e6, "block": "8 frames per qk_process_samples_device call, one row released per block (qk_graph_mark behind the block, qk_get_graph while the next block runs)",
           "row_finite": bool(row is not None and np.isfinite(row).all()),
           "roofline": {"kernel": "k_fft_cols + k_fft_rows, per frame (a launch pair transforms 8 frames; HIP events on their stream)", "bound": "hbm",
                        "achieved": k_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k_gbs / HBM_PEAK_GBS,
                        "avg_frame_kernels_ms": k_ms, "traffic": traffic, "traffic_source": source,
                        "algorithmic_bytes_per_frame": N * 16.0,
                        "whole_leg": {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                                      "note": "frames/s x 16 B x 2^20 over wall time, release and host calls included"}}}
    if with_cpu:
        res["cpu_baseline"] = cpu_fft_baseline(N)
    return res


def config1_leg(local, with_cpu):
    """BASELINE configs[0]: 1 USB receiver on a 192 ksps stream, 960-sample blocks (the reference's 5 ms poll), 4096-point
    graph FFT: the engine fed host blocks one after the other (launch-bound at this size) beside the CPU port on one thread
    (the reference runs this on its single sound thread)."""
    from quisk_amd import Engine, rx_filter
    from quisk_amd.synth import make_iq
    fs, blk, nblk = 192000, 960, 1000
    tune, mode = int(0.1 * fs), 3
    x = make_iq(blk * nblk, fs, [tune], [mode], 1001)
    eng = Engine(fs, n_receivers=1, max_block=blk, fft_size=4096, data_width=4096, device=local)
    fi, fq, bw, off = rx_filter(fs, mode)
    eng.set_mode(0, mode); eng.set_tune(0, tune); eng.set_filters(0, fi, fq, bw, off)
    out = np.empty(eng.out_cap, dtype=np.complex128)
    for i in range(20):
        eng.process(x[i * blk:(i + 1) * blk])
    eng.sync()
    t0 = time.perf_counter()
    frames = 0
    for i in range(nblk):
        eng.process(x[i * blk:(i + 1) * blk])
        eng.L.qk_pull_audio(eng.h, 0, out.ctypes.data, eng.out_cap)     # audio back on the host every block, as the sound loop needs it
        if i % 10 == 9:
            frames += eng.get_graph(1.0, 0.0)[0]
    eng.sync()
    dt = time.perf_counter() - t0
    eng.close()
    res = {"workload": "BASELINE configs[0]: 1 USB receiver, 192 ksps, 960-sample host blocks, audio pulled every block, 4096-point graph",
           "msamples_per_s": blk * nblk / dt / 1e6, "realtime_factor": blk * nblk / dt / fs, "ms_per_block": dt / nblk * 1e3,
           "fft_frames": frames, "note": "short-block mode (max_block <= 4096): one stream, front end -> one fused chain launch -> AGC gain + apply in one launch; input written by the host into device memory, audio rows in pinned host memory; tail copies and spectrum frames behind the audio event"}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _bind as B
        from _rxsetup import make_oracle_rx
        o = make_oracle_rx(fs, mode, tune)
        g = B.OracleGraph(4096, 4096, fs)
        t0 = time.perf_counter()
        for i in range(nblk):
            o.process(x[i * blk:(i + 1) * blk])
        for i in range(0, blk * nblk - 4095, 4096):
            g.add_frames(x[i:i + 4096])
        g.release()
        cdt = time.perf_counter() - t0
        o.close(); g.close()
        res["cpu_baseline"] = {"value": blk * nblk / cdt / 1e6, "unit": "input Msamples/s at 1 receiver", "cores": 1, "kind": "port",
                               "realtime_factor": blk * nblk / cdt / fs,
                               "sample": "%d blocks of 960 samples (5 s of signal) + %d graph frames, one thread" % (nblk, blk * nblk // 4096)}
    return res


def main():
    a = parse()
    global PREROLL_MS
    PREROLL_MS = a.preroll_ms
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world
    if a.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (a.gpus, world))

    from quisk_amd.engine import KT_FRONTEND
    from quisk_amd.synth import make_iq, tune_offsets, mode_mix
    from quisk_amd.shard import receiver_slice, max_over_ranks

    fs, nrx, block = a.rate, a.receivers, a.block
    total_rx = nrx * world
    all_tunes = tune_offsets(total_rx, fs)
    all_modes = mode_mix(total_rx, a.mix)
    lo, hi = receiver_slice(total_rx, world, rank)
    tunes, modes = all_tunes[lo:hi], all_modes[lo:hi]
    eng = make_engine(fs, nrx, block, local, tunes, modes, fp32=a.fp32)

    # synthetic stream resident in HBM: NB distinct blocks (the base block rotated by a different phase each),
    # 256 MiB in total so that successive steps do not re-read a cache-resident input
    nb = max(2, min(16, (256 << 20) // (block * 16)))
    base = make_iq(block, fs, all_tunes[:: max(1, total_rx // 16)][:16], all_modes[:: max(1, total_rx // 16)][:16], 1002)
    host = np.empty((nb, block), dtype=np.complex128)
    for b in range(nb):
        host[b] = np.round(base * np.exp(1j * 0.37 * b))
    dev = torch.from_numpy(host.view(np.float64).reshape(nb, block, 2)).cuda()
    ptrs = [dev[b].data_ptr() for b in range(nb)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comm_bad, comm_err = set(), []

    def init_comm(e):
        """the library's own RCCL communicator (qk_comm_init): rank 0 makes the id, the launcher's process group only carries it"""
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            try:
                uid.copy_(torch.frombuffer(bytearray(e.comm_unique_id()), dtype=torch.uint8))
            except Exception as ex:      # (an all-zero id tells the other ranks)
                comm_err.append(str(ex))
        dist.broadcast(uid, 0)
        ok = torch.ones(1, dtype=torch.int32, device="cuda")
        try:
            if not bool(uid.any().item()):
                raise RuntimeError("no unique id from rank 0")
            e.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
        except Exception as ex:          # the gather is reported, not measured as `value`: a rank without a communicator
            ok.zero_()                   # must not cost the run its headline -- every rank then leaves the gather out
            comm_err.append(str(ex))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm_bad.add(id(e))

    def gather_outputs(e, n_local):
        """The only collective of the path: the last block's audio of every receiver to rank 0, over RCCL from inside libqkengine
        (qk_gather_audio: ncclSend / ncclRecv in one group, straight out of the engine's audio rows)."""
        if id(e) in comm_bad:
            return {"backend": "rccl-c", "error": "qk_comm_init failed on a rank: " + (comm_err[0] if comm_err else "see the other ranks")}
        _, stride = e.audio_device()
        dst = torch.empty((world, n_local, stride, 2), dtype=torch.float64, device="cuda") if rank == 0 else None
        cnt = torch.zeros((world, n_local), dtype=torch.int32, device="cuda") if rank == 0 else None
        barrier()
        g0 = time.perf_counter()
        e.gather_audio(0, dst.data_ptr() if rank == 0 else 0, cnt.data_ptr() if rank == 0 else 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert int(cnt.min().item()) > 0 and bool(torch.isfinite(dst[:, :, :int(cnt.min().item())]).all().item())
        return {"ms": ms, "bytes": int(n_local * world * stride * 16), "receivers": n_local * world, "backend": "rccl-c"}

    if world > 1:
        init_comm(eng)

    # The headline leg runs twice: once straight after set-up, when the GPU has been idle (its clocks take some 20 ms of
    # load to come up: a 20-step timed region lies wholly inside that ramp), and again -- the figure reported as `value` --
    # behind a pre-roll of its own blocks (PREROLL_MS, untimed), i.e. at the clocks a receiver that runs for more than a few
    # milliseconds sees.  Both are W warm-up steps followed by exactly K timed steps between barriers.
    dt_idle, _ = timed_leg(eng, ptrs, block, a.steps, a.warmup, barrier, (KT_FRONTEND,), preroll_ms=0.0)
    dt_idle = max_over_ranks(dt_idle, torch.device("cuda", local))
    res = {}
    later = []

    # ---- the north-star slice: configs[3] per GPU ----
    if not a.no_config4 and block >= (1 << 20):
        fs4, nrx4 = 61440000, 128
        t4 = tune_offsets(nrx4 * world, fs4)
        m4 = mode_mix(nrx4 * world, "mixed")
        lo4, hi4 = receiver_slice(nrx4 * world, world, rank)
        eng4 = make_engine(fs4, nrx4, block, local, t4[lo4:hi4], m4[lo4:hi4], fp32=a.fp32)
        k4 = min(a.steps, 100)
        dt4, st4 = timed_leg(eng4, ptrs, block, k4, min(a.warmup, 10), barrier, (KT_FRONTEND,))
        dt4 = max_over_ranks(dt4, torch.device("cuda", local))
        v4 = k4 * block * world / dt4 / 1e6
        leg = {"workload": "per-GPU share of BASELINE configs[3]: %d mixed SSB/AM/FM receivers per GPU at 61.44 Msps, "
                           "2^20-sample blocks, AGC" % nrx4,
               "value": v4, "unit": "Msamples/s", "steps": k4, "ms_per_step": dt4 / k4 * 1e3, "n_gpus": world,
               "receivers_total": nrx4 * world, "realtime_factor": v4 * 1e6 / world / fs4,
               "roofline": frontend_roofline(st4, fs4, block, nrx4, dt4 / k4 * 1e3, a.fp32, "r03_pmc_frontend_c4.json", "r03_kernel_stats_default_bench.csv")}
        if world > 1:
            init_comm(eng4)
            leg["gather"] = gather_outputs(eng4, nrx4)
        later.append(eng4)       # (closed behind the headline leg: freeing its buffers would idle the GPU in front of it)
        res["config4_slice"] = leg

    # ---- configs[4] per GPU: the same slice in the engine's fp32 mode with the 2^20-point display FFT running beside it ----
    if not a.no_config5 and block >= (1 << 20):
        fs5, nrx5, N5 = 61440000, 128, 1 << 20
        t5 = tune_offsets(nrx5 * world, fs5)
        m5 = mode_mix(nrx5 * world, "mixed")
        lo5, hi5 = receiver_slice(nrx5 * world, world, rank)
        eng5 = make_engine(fs5, nrx5, block, local, t5[lo5:hi5], m5[lo5:hi5], fp32=True, fft_size=N5, data_width=4096)
        k5 = min(a.steps, 100)
        dt5, st5 = timed_leg(eng5, ptrs, block, k5, min(a.warmup, 10), barrier, (KT_FRONTEND,))
        dt5 = max_over_ranks(dt5, torch.device("cuda", local))
        frames5, row5, _ = eng5.get_graph(1.0, 0.0)
        v5 = k5 * block * world / dt5 / 1e6
        res["config5_slice"] = {
            "workload": "per-GPU share of BASELINE configs[4]: %d mixed receivers per GPU at 61.44 Msps in fp32 mode (front end, tiled FIRs, "
                        "FFT in float; phasor, detectors, AGC in double) with a 2^20-point windowed FFT of every block beside them" % nrx5,
            "dtype": "f32", "value": v5, "unit": "Msamples/s", "steps": k5, "ms_per_step": dt5 / k5 * 1e3, "n_gpus": world,
            "receivers_total": nrx5 * world, "realtime_factor": v5 * 1e6 / world / fs5,
            "fft_frames_per_s": float(st5.fft_frames) / dt5 if dt5 > 0 else 0.0, "frames_averaged_at_the_end": int(frames5),
            "row_finite": bool(row5 is not None and np.isfinite(row5).all()),
            "tolerance": "tests/test_gpu_fp32.py: audio in front of the AGC <= 1e-5 relative RMS (floor 1e-8 of the stream's peak), "
                         "spectrum <= 0.01 dB, against the fp64 oracle",
            "roofline": frontend_roofline(st5, fs5, block, nrx5, dt5 / k5 * 1e3, True, "r03_pmc_frontend_c5.json", "r03_kernel_stats_default_bench.csv")}
        later.append(eng5)

    # ---- the whole of configs[3] on ONE GPU: 1024 mixed receivers at 61.44 Msps + the fp64 2^20-point FFT of every block ----
    if not a.no_one_gpu and not a.no_config4 and world == 1 and block >= (1 << 20):
        fs6, nrx6, N6 = 61440000, 1024, 1 << 20
        eng6 = make_engine(fs6, nrx6, block, local, tune_offsets(nrx6, fs6), mode_mix(nrx6, "mixed"), fp32=a.fp32, fft_size=N6, data_width=4096)
        k6 = min(a.steps, 20)
        dt6, st6 = timed_leg(eng6, ptrs, block, k6, 2, barrier, (KT_FRONTEND,))
        frames6, row6, _ = eng6.get_graph(1.0, 0.0)
        v6 = k6 * block / dt6 / 1e6
        res["config4_one_gpu"] = {
            "workload": "BASELINE configs[3] whole on ONE GPU: %d mixed SSB/AM/FM receivers at 61.44 Msps, 2^20-sample blocks, AGC, "
                        "with a 2^20-point fp64 windowed FFT of every block beside them" % nrx6,
            "value": v6, "unit": "Msamples/s", "steps": k6, "ms_per_step": dt6 / k6 * 1e3, "n_gpus": 1, "receivers_total": nrx6,
            "realtime_factor": v6 * 1e6 / fs6, "fits_one_gpu_in_real_time": bool(v6 * 1e6 / fs6 >= 1.0),
            "fft_frames_per_s": float(st6.fft_frames) / dt6 if dt6 > 0 else 0.0,
            "row_finite": bool(row6 is not None and np.isfinite(row6).all()),
            "roofline": frontend_roofline(st6, fs6, block, nrx6, dt6 / k6 * 1e3, a.fp32, "r03_pmc_frontend_c4.json")}
        eng6.close()

    dt, st = timed_leg(eng, ptrs, block, a.steps, a.warmup, barrier, (KT_FRONTEND,))
    dt = max_over_ranks(dt, torch.device("cuda", local))
    gather = gather_outputs(eng, nrx) if world > 1 else None

    value = a.steps * block * world / dt / 1e6
    step_ms = dt / a.steps * 1e3
    res0 = res
    res = {
        "metric": "input Msamples/s processed (whole node) at %d receivers per GPU" % nrx,
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.fp32 else "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d %s receivers per GPU on a %.3f Msps complex-double stream, "
                               "mixer + cascaded decimating FIRs + demod + AGC" % (nrx, "SSB" if a.mix == "ssb" else "mixed SSB/AM/FM", fs / 1e6),
                   "block_samples": block, "receivers_total": total_rx, "sample_rate": fs, "sharding": "receivers"},
        "realtime_factor": value * 1e6 / world / fs,
        "receiver_msamples_per_s": value * nrx,
        "roofline": frontend_roofline(st, fs, block, nrx, step_ms, a.fp32, "r03_pmc_frontend_c2.json", "r03_kernel_stats_default_bench.csv"),
    }
    res["after_idle"] = {"ms_per_step": dt_idle / a.steps * 1e3, "value": a.steps * block * world / dt_idle / 1e6,
                         "note": "the same leg run first, straight after set-up and without the pre-roll: the GPU's clocks are still coming up from idle"}
    res["preroll_ms"] = PREROLL_MS
    res["preroll_note"] = ("clock conditioning outside the measurement: every leg runs its own blocks untimed for this long in front of its W warm-up "
                           "steps (the GPU's clocks need ~20 ms of load to come up from idle; a receiver runs for hours); the timed region is "
                           "still exactly K steps between barriers; `after_idle` is the figure without it")
    res.update(res0)
    if gather is not None:
        res["gather"] = gather
    eng.close()
    for e_ in later:
        e_.close()

    with_cpu = rank == 0 and not a.no_cpu_baseline and world == 1
    if rank == 0 and not a.no_spectrum:
        sp = spectrum_leg(local, ptrs, block, with_cpu)
        if sp is not None:
            res["spectrum"] = sp
    if rank == 0 and not a.no_config1 and world == 1:
        res["config1"] = config1_leg(local, with_cpu)
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline(fs, modes, tunes, a.cpu_samples, 1002)
        res["cpu_baseline"]["reference_calibration"] = reference_calibration()
        if "config4_slice" in res:
            fs4, nrx4 = 61440000, 128
            c4 = cpu_baseline(fs4, mode_mix(nrx4, "mixed"), tune_offsets(nrx4, fs4), 1 << 18, 1004)
            res["config4_slice"]["cpu_baseline"] = c4
            if "config5_slice" in res:
                fft = res.get("spectrum", {}).get("cpu_baseline")
                c5 = dict(c4)
                c5["sample"] += "; the port computes in fp64 (the reference has no fp32 mode)"
                if fft:     # one 2^20-point frame per 2^20-sample block, after the receivers, on the same cores
                    t_blk = (1 << 20) / (c4["value"] * 1e6) + 1.0 / fft["value"]
                    c5["value"] = (1 << 20) / t_blk / 1e6
                    c5["realtime_factor"] = c5["value"] * 1e6 / fs4
                    c5["sample"] += " + the oracle's 2^20-point FFT of every block (%.1f frames/s on %d cores)" % (fft["value"], fft["cores"])
                res["config5_slice"]["cpu_baseline"] = c5
            if "config4_one_gpu" in res:
                c6 = dict(c4)
                c6["value"] = c4["value"] * nrx4 / 1024.0
                c6["realtime_factor"] = c6["value"] * 1e6 / fs4
                c6["unit"] = "input Msamples/s at 1024 receivers"
                c6["sample"] = "scaled from the 128-receiver run above (receivers are independent: 1024 take 8 times as long); FFT not included"
                res["config4_one_gpu"]["cpu_baseline"] = c6
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
