"""The N > 1 path on CPUs: two processes over gloo shard six receivers (uneven: 3 + 3, then 5 = 3 + 2), each
computes its receivers' audio with the oracle, rank 0 gathers; the result must equal a single-process run."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quisk_amd.shard import gather_audio, max_over_ranks, receiver_slice


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _audio_for(rx_ids, fs, tunes, modes, x, block):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _rxsetup import make_oracle_rx
    rows, counts = [], []
    for r in rx_ids:
        o = make_oracle_rx(fs, int(modes[r]), int(tunes[r]))
        out = o.process_stream(x, block)
        o.close()
        rows.append(out); counts.append(len(out))
    stride = max(counts)
    arr = np.zeros((len(rx_ids), stride, 2))
    for i, out in enumerate(rows):
        arr[i, :len(out), 0] = out.real
        arr[i, :len(out), 1] = out.imag
    return torch.from_numpy(arr), torch.tensor(counts, dtype=torch.int64)


def _worker(rank, world, port, total_rx, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quisk_amd.synth import make_iq, tune_offsets, mode_mix
    fs, block = 192000, 9600
    tunes, modes = tune_offsets(total_rx, fs), mode_mix(total_rx, "mixed")
    x = make_iq(block * 4, fs, tunes, modes, 21)
    lo, hi = receiver_slice(total_rx, world, rank)
    local, counts = _audio_for(range(lo, hi), fs, tunes, modes, x, block)
    audio, cnt = gather_audio(local, counts, dst=0)
    t = max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert t == float(world)
    if rank == 0:
        ref, rcnt = _audio_for(range(total_rx), fs, tunes, modes, x, block)
        q.put((bool(torch.equal(audio, ref)), bool(torch.equal(cnt, rcnt)), tuple(audio.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(total_rx):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total_rx, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_receiver_slice_partitions_exactly():
    for total in (1, 5, 64, 1024, 1000):
        for world in (1, 2, 3, 8):
            got = [receiver_slice(total, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_even_and_uneven():
    for total in (6, 5):
        same_audio, same_counts, shape = _run(total)
        assert same_audio and same_counts and shape[0] == total
