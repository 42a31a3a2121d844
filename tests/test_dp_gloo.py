"""CPU, world_size 2 over gloo: the data-parallel contract of the train step.

Each rank takes a contiguous shard of the global batch, computes the shard's mean-reduced CTC gradient (here with the
CPU oracle standing in for the HIP engine), the flat gradient buffer is sum-all-reduced ONCE and the 1/world factor is
folded into the optimizer's scale.  The result must equal the single-process gradient of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import w2v2_ref as R
from ssak_amd.data import shard_batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    cfg = R.W2V2Config.tiny().deterministic()
    p = R.init_params(cfg, 7)
    rng = np.random.default_rng(0)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(6000).astype(np.float32) for _ in range(4)])
    labels = R.pad_labels([[3, 4, 5], [6, 7, 8], [9, 10, 11], [12, 13, 14]])  # equal target lengths
    return cfg, p, x, labels


def _flat(grads, names):
    return torch.cat([grads[n].reshape(-1) for n in names])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg, p, x, labels = _problem()
    mine = shard_batch(list(range(4)), rank, world)
    loss, _, g = R.loss_and_grads(p, cfg, torch.tensor(x[mine]), None, torch.tensor(labels[mine]))
    names = R.trainable_names(cfg)
    flat = _flat(g, names)
    dist.all_reduce(flat)            # ONE collective for the whole gradient
    flat *= 1.0 / world              # folded into the optimizer's grad_scale in ssak_amd.trainer
    lt = torch.tensor([loss.item()], dtype=torch.float64)
    dist.all_reduce(lt)
    if rank == 0:
        torch.save({"flat": flat, "loss": lt / world}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_equals_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg, p, x, labels = _problem()
    loss, _, g = R.loss_and_grads(p, cfg, torch.tensor(x), None, torch.tensor(labels))
    ref = _flat(g, R.trainable_names(cfg))
    assert abs(got["loss"].item() - loss.item()) < 1e-5
    assert (got["flat"] - ref).abs().max().item() < 1e-5 * ref.abs().max().item() + 1e-7
