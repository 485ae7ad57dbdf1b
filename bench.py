#!/usr/bin/env python3
"""bench.py -- throughput of the receive-DSP hot path on MI355X.

Workload (BASELINE.json configs[1]): 64 SSB receivers per GPU on one 1.536 Msps complex-double I/Q stream,
cascaded decimating FIRs, SSB demodulation, AGC, fp64.  A "step" is one pass of the whole receive chain
(quisk_process_samples for every receiver) over one block of `--block` synthetic I/Q samples that are already
resident in HBM.  value = input samples processed per second (summed over ranks) / 1e6.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): receivers are sharded, 64 per rank, every rank reads
the same I/Q stream (weak scaling, no data-path collective).  The audio gather over RCCL that the north star
asks for is exercised once after the timed region and timed separately (`gather_ms`).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FE_FLOPS_PER_SAMPLE = {0: 12.0, 1: 12 + 34.0, 2: 12 + 34 * 1.5, 3: 12 + 34 * 1.75, 4: 12 + 34 * 1.875, 5: 12 + 34 * 1.9375}
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak: half the 157.3 TF FP32 vector rate of MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 200 blocks of 2^20 samples = 0.09 s of GPU time: fill and drain of the
    ap.add_argument("--warmup", type=int, default=10)     # three-deep pipeline are 1 % of it (they are 4 % of 40 steps)
    ap.add_argument("--block", type=int, default=1 << 20, help="I/Q samples per step")
    ap.add_argument("--receivers", type=int, default=64, help="receivers per GPU")
    ap.add_argument("--rate", type=int, default=1536000)
    ap.add_argument("--mix", default="ssb", choices=["ssb", "mixed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spectrum", action="store_true", help="skip the configs[2] spectrum leg")
    ap.add_argument("--cpu-samples", type=int, default=1 << 21)
    return ap.parse_args()


def cpu_baseline(fs, modes, tunes, n_samples, seed):
    """The oracle (CPU restatement of the reference's per-sample loops, kind "port") on the host cores:
    one receiver per thread, blocks of 65536 like config 2."""
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
            "kind": "port",
            "sample": "%d receivers x %d I/Q samples (%.2f s of signal) in 65536-sample blocks, one receiver per thread"
                      % (len(modes), n_samples, n_samples / fs)}


def spectrum_leg(local, dev_block_ptrs, block):
    """BASELINE configs[2]: 2^20-point periodic-Hann windowed FFT at 61.44 Msps, magnitude average of 8 frames
    released as a 4096-point dB row (get_graph).  Frames come from the blocks already resident in HBM.
    Reported beside the headline number: frames/s and the algorithmic HBM rate (16 B per sample read once)."""
    import torch
    from quisk_amd import Engine
    N = 1 << 20
    if block < N:
        return None
    eng = Engine(61440000, n_receivers=0, max_block=N, fft_size=N, data_width=4096, device=local)
    row = None
    for i in range(8):
        eng.process_device(dev_block_ptrs[i % len(dev_block_ptrs)], N)
    _, row, _ = eng.get_graph(1.0, 0.0)
    torch.cuda.synchronize()
    groups, t0 = 6, time.perf_counter()
    for g in range(groups):
        for i in range(8):
            eng.process_device(dev_block_ptrs[i % len(dev_block_ptrs)], N)
        _, row, _ = eng.get_graph(1.0, 0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.close()
    fps = groups * 8 / dt
    gbs = fps * N * 16.0 / 1e9
    return {"workload": "BASELINE configs[2]: 2^20-point windowed FFT at 61.44 Msps, 8-frame magnitude average, 4096-point dB row",
            "frames_per_s": fps, "realtime_frames_per_s": 61440000.0 / N, "msamples_per_s": fps * N / 1e6,
            "row_finite": bool(row is not None and np.isfinite(row).all()),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_frame": N * 16.0}}


def main():
    a = parse()
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

    from quisk_amd import Engine
    from quisk_amd.engine import KT_FRONTEND
    from quisk_amd.synth import make_iq, tune_offsets, mode_mix
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _rxsetup import rx_filter

    fs, nrx, block = a.rate, a.receivers, a.block
    total_rx = nrx * world
    all_tunes = tune_offsets(total_rx, fs)
    all_modes = mode_mix(total_rx, a.mix)
    from quisk_amd.shard import receiver_slice, gather_audio, max_over_ranks
    lo, hi = receiver_slice(total_rx, world, rank)
    tunes, modes = all_tunes[lo:hi], all_modes[lo:hi]

    eng = Engine(fs, n_receivers=nrx, max_block=block, device=local)
    for r in range(nrx):
        fi, fq, bw, off = rx_filter(fs, int(modes[r]))
        eng.set_mode(r, int(modes[r]))
        eng.set_tune(r, int(tunes[r]))
        eng.set_filters(r, fi, fq, bw, off)

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

    for i in range(a.warmup):
        eng.process_device(ptrs[i % nb], block)
    eng.sync()
    eng.reset_stats()
    eng.set_timing(True, groups=(KT_FRONTEND,))      # HIP events around the front-end kernel only
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.process_device(ptrs[i % nb], block)
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    eng.set_timing(False)
    dt = max_over_ranks(dt, torch.device("cuda", local))

    # RCCL gather of the audio outputs (outside the timed region)
    gather_ms = None
    if world > 1:
        counts = torch.tensor([eng.L.qk_audio_count(eng.h, r) for r in range(nrx)], dtype=torch.int64, device="cuda")
        stride = int(counts.max().item())
        send = torch.empty((nrx, stride, 2), dtype=torch.float64, device="cuda")
        eng.export_audio_device(send.data_ptr(), stride)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        audio, cnt = gather_audio(send, counts, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert audio.shape[0] == total_rx and int(cnt.min().item()) > 0

    from quisk_amd import plan_decimation
    _, d2, _, _ = plan_decimation(fs)
    S = min(max(d2 - 1, 0), 5)
    fe_ms = st.kernel_ms_sum[KT_FRONTEND] / max(1, st.kernel_launches[KT_FRONTEND])
    fe_flops = FE_FLOPS_PER_SAMPLE[S] * block * nrx
    achieved_tf = fe_flops / (fe_ms * 1e-3) / 1e12 if fe_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_frontend.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    value = a.steps * block * world / dt / 1e6
    res = {
        "metric": "input Msamples/s processed (whole node) at %d receivers per GPU" % nrx,
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d %s receivers per GPU on a %.3f Msps complex-double stream, "
                               "mixer + cascaded decimating FIRs + demod + AGC" % (nrx, "SSB" if a.mix == "ssb" else "mixed SSB/AM/FM", fs / 1e6),
                   "block_samples": block, "receivers_total": total_rx, "sample_rate": fs, "sharding": "receivers"},
        "realtime_factor": value * 1e6 / world / fs,
        "receiver_msamples_per_s": value * nrx,
        # "mfma": the compute roof.  The kernel is FP64 FMA work issued on the vector pipe; on MI355X the dense FP64
        # MFMA peak and the FP64 vector peak are the same 78.6 TFLOP/s, so one number prices both.
        "roofline": {"kernel": "k_frontend_wave<S=%d> (mixer + %d fused half-band decimators)" % (S, S),
                     "bound": "mfma", "pipe": "fp64 vector FMA", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved_tf / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                     "avg_launch_ms": fe_ms, "launches": int(st.kernel_launches[KT_FRONTEND]),
                     "algorithmic_flop_per_launch": fe_flops,
                     "hbm": {"achieved": (16.0 * block + 16.0 * (block >> S) * nrx) / (fe_ms * 1e-3) / 1e9 if fe_ms > 0 else 0.0,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s"},
                     "share_of_step": fe_ms / (dt / a.steps * 1e3) if dt > 0 else None},
    }
    if gather_ms is not None:
        res["gather_ms"] = gather_ms
    if rank == 0 and not a.no_spectrum:
        sp = spectrum_leg(local, ptrs, block)
        if sp is not None:
            res["spectrum"] = sp
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(fs, modes, tunes, a.cpu_samples, 1002)
    eng.close()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
