"""Training loop -- host-side mirror of the reference's train.py:24-207 on top of libdesire_hip.so.

Kept from the reference: the argparse flags and defaults of train.py:28-88, the epoch loop with
`lr = learning_rate * decay_rate ** epoch` (train.py:122-126), `reset_batch_pointer()` per epoch, the save cadence
`(epoch * num_batches + batch) % save_every == 0 and > 0` (train.py:197-206), the per-batch log line
(train.py:185-193) and `save/config.pkl` (train.py:100-101).

Different on purpose: the reference fetches only `cost` (train.py:181, its Adam op is never run) one window at a
time; here every batch is ONE forward + backward + clip + Adam step over all `batch_size` windows
(`DESIREModel.train_step`), the loader window is `seq_length + pred_length` frames split into past / future (the
reference has one length and feeds the window shifted by a frame as "target"), and checkpoints are named fp32 `.npz`
files (formats.save_weights) instead of TF checkpoints.  With torch.distributed initialised every rank takes its block
of the batch's windows and gradients are averaged with one flat all-reduce (dist.allreduce_mean_).

    python -m desire_amd.train --data_dir data/ --batch_size 16 --max_num_obj 32 --d_dim 128 --num_epochs 1
"""
from __future__ import annotations

import argparse
import os
import pickle
import sys
import time
from typing import Callable, List, Sequence, Tuple

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="DESIRE training on MI355X (flags of the reference's train.py:28-88)")
    p.add_argument("--rnn_size", type=int, default=512)
    p.add_argument("--num_layers", type=int, default=1)
    p.add_argument("--model", type=str, default="gru")
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--seq_length", type=int, default=8)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--save_every", type=int, default=400)
    p.add_argument("--grad_clip", type=float, default=10.0)
    p.add_argument("--learning_rate", type=float, default=0.005)
    p.add_argument("--decay_rate", type=float, default=0.95)
    p.add_argument("--keep_prob", type=float, default=0.8)
    p.add_argument("--embedding_size", type=int, default=64)
    p.add_argument("--neighborhood_size", type=int, default=32)
    p.add_argument("--grid_size", type=int, default=4)
    p.add_argument("--max_num_obj", type=int, default=60)
    p.add_argument("--leave_dataset", type=int, default=5)
    p.add_argument("--latent_size", type=int, default=128)
    p.add_argument("--e_dim", type=int, default=256)
    p.add_argument("--d_dim", type=int, default=16)
    p.add_argument("--stride", type=int, default=1)
    # ---- not in the reference ----
    p.add_argument("--pred_length", type=int, default=None, help="future frames (default: seq_length)")
    p.add_argument("--num_samples", type=int, default=20, help="K futures per agent")
    p.add_argument("--data_dir", type=str, default="data/")
    p.add_argument("--traj_bin", type=str, default=None, help="DSRTRJ1 container instead of CSV/cpkl")
    p.add_argument("--save_dir", type=str, default="save")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many optimiser steps (0 = all epochs)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--fix_id0", action="store_true", help="keep SDD track id 0 (the reference drops it)")
    p.add_argument("--ioc_iters", type=int, default=1, help="IOC refinement passes")
    p.add_argument("--bf16", type=str, default="", choices=["", "f32", "x3", "split"],
                   help="matrix operands of the training step: fp32 (default) or x3 = split-bf16 (hi + lo, three bf16 MFMAs per product) in the IOC "
                        "forward / BPTT, the weight-gradient reductions and the large data-gradient convolutions; gradients stay within the fp32 "
                        "path's tolerance of float64 autograd")
    p.add_argument("--two_piece_forward", action="store_true",
                   help="with --bf16 x3: two-piece operands in the forward pass's sample generation too (DESIRE_FLAG_TRAIN_FWD_3P: ~4 %% faster "
                        "steps, gradients within 5e-4 instead of 2e-4 of float64 autograd)")
    p.add_argument("--skip_padding", action="store_true",
                   help="(the default since round 6; kept so that older command lines still parse) DESIRE_FLAG_COMPACT_ROWS | DESIRE_FLAG_COMPACT_IOC: "
                        "run the step on the agents that are present only -- the per-row stages on present rows, the IOC on slot classes "
                        "(include/desire_hip.h).  On SDD windows most of max_num_obj is padding (utils/data_loader.py:209-229): 2.3x faster steps on "
                        "bookstore/video6 at max_num_obj 32; gradients equal the padded step's to fp32 reduction noise")
    p.add_argument("--keep_padding", action="store_true",
                   help="run every one of the max_num_obj slots of every window like the reference does (it masks id-0 objects in the cost only, "
                        "model/model.py:351-366): the opt-out of the default above.  The compacted step reads the present-agent counts back once per "
                        "step (one host wait on an event, include/desire_hip.h)")
    p.add_argument("--head_loss_weight", type=float, default=0.0,
                   help="weight of the reference's own loss for the 5-wide Gaussian output layer (model/model.py:494-550: -log N(next position | "
                        "mux, muy, sx, sy, rho), teacher-forced over the observed frames) added to the training loss; > 0 trains gauss_head/w|b "
                        "(and the X encoder through it), i.e. what sample(mode='rollout') reads")
    p.add_argument("--prefetch", type=int, default=2,
                   help="batches the loader thread keeps in flight (pinned staging -> copy stream -> device; desire_amd/prefetch.py); 0 = the "
                        "reference's serial loop: next_batch, copy, step, read the loss, every step (train.py:131-183)")
    p.add_argument("--report_ade", action="store_true",
                   help="after every epoch: ADE / FDE (mean-of-K and best-of-K, normalised units) of PRIOR samples on the epoch's last batch")
    return p


def lr_at_epoch(args, epoch: int) -> float:
    """train.py:122-126."""
    return float(args.learning_rate) * float(args.decay_rate) ** int(epoch)


def should_save(epoch: int, batch: int, num_batches: int, save_every: int) -> bool:
    """train.py:197-199."""
    step = epoch * num_batches + batch
    return step % save_every == 0 and step > 0


def split_windows(xval: Sequence[np.ndarray], t_obs: int) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Loader windows of seq_length + pred_length frames -> (past [T_obs, MNO, 3], future [T_pred, MNO, 3])."""
    past = [np.asarray(x)[:t_obs] for x in xval]
    fut = [np.asarray(x)[t_obs:] for x in xval]
    return past, fut


def train(args, data_loader=None, model=None, log: Callable[[str], None] = print) -> List[float]:
    """Runs the loop; returns the per-step losses.  `data_loader` / `model` may be injected (tests)."""
    from .data_loader import DataLoader
    from .dist import shard_batch
    from .model import DESIREModel
    import torch.distributed as dist

    if args.pred_length is None:
        args.pred_length = args.seq_length
    t_obs, t_pred = int(args.seq_length), int(args.pred_length)
    if data_loader is None:
        data_loader = DataLoader(args.batch_size, t_obs + t_pred, args.max_num_obj, args.leave_dataset, preprocess=False,
                                 data_dir=args.data_dir, traj_bin=args.traj_bin, fix_id0=args.fix_id0)
    if data_loader.seq_length != t_obs + t_pred:
        raise ValueError("the loader window must be seq_length + pred_length frames")
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    if args.batch_size < world:
        raise ValueError("--batch_size %d is smaller than the %d ranks: every rank needs at least one window per step" % (args.batch_size, world))
    # next_batch advances its pointer by random.randint (utils/data_loader.py:235-238): the SAME seed on every rank, so that all
    # ranks draw the same global batch and shard_batch partitions it (and --seed makes a run reproducible)
    import random
    random.seed(int(args.seed))
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        with open(os.path.join(args.save_dir, "config.pkl"), "wb") as fh:
            pickle.dump(args, fh)
    if model is None:
        model = DESIREModel(args, seed=args.seed)
    losses: List[float] = []
    # the overlapped schedule needs the fast loader's next_batch_into; a reference-shaped loader (utils/data_loader.py's API: next_batch only) gets
    # the reference's serial loop instead of an AttributeError from the feeder thread
    if int(getattr(args, "prefetch", 0) or 0) > 0 and hasattr(data_loader, "next_batch_into"):
        return _train_overlapped(args, data_loader, model, log, rank, world, t_obs, t_pred, losses)
    steps = 0
    for epoch in range(args.num_epochs):
        model.learning_rate = lr_at_epoch(args, epoch)
        data_loader.reset_batch_pointer()
        for batch in range(data_loader.num_batches):
            start = time.time()
            xval, _, _ = data_loader.next_batch()
            past, fut = split_windows(shard_batch(xval, rank, world), t_obs)
            terms = model.train_step(past, fut, seed=args.seed + steps * world + rank)
            losses.append(terms["loss"])
            steps += 1
            if rank == 0:
                log("{}/{} (epoch {}), train_loss = {:.3f}, time/batch = {:.3f}".format(
                    epoch * data_loader.num_batches + batch, args.num_epochs * data_loader.num_batches, epoch,
                    terms["loss"], time.time() - start))
                sys.stdout.flush()
            if rank == 0 and should_save(epoch, batch, data_loader.num_batches, args.save_every):
                path = os.path.join(args.save_dir, "social_model-%d.npz" % (epoch * data_loader.num_batches + batch))
                model.save(path)
                log("model saved to {}".format(path))
            if args.max_steps and steps >= args.max_steps:
                return losses
        if getattr(args, "report_ade", False) and data_loader.num_batches > 0:
            _report_ade(args, model, past, fut, epoch, rank, log)
    return losses


def _report_ade(args, model, past, fut, epoch, rank, log) -> None:
    """The evaluation harness the reference never had: prior sampling (no future given) on this rank's share of the
    epoch's last batch, masked like the loss (present at the last observed frame and in every future frame).  past / fut: lists of
    loader windows or device tensors [n, T, mno, 3]."""
    import torch
    if torch.is_tensor(past):
        Y, _ = model.forward_device(past, None, seed=args.seed)
        pw, fw = past.cpu().numpy(), fut.cpu().numpy()
    else:
        Y, _ = model.forward(past, None, seed=args.seed)
        pw, fw = np.stack([np.asarray(p) for p in past]), np.stack([np.asarray(f) for f in fut])
    ev = model.evaluate(Y, fut)
    there = np.zeros(ev.shape[0], bool)
    m = pw.shape[2]
    there.reshape(pw.shape[0], -1)[:, :m] = (pw[:, -1, :, 0] != 0) & (fw[:, :, :, 0] != 0).all(1)
    if there.any():
        e = ev[there].mean(0)
        log("epoch {} rank {}: ADE/FDE mean-of-K = {:.5f} / {:.5f}, best-of-K = {:.5f} / {:.5f} ({} agents)".format(
            epoch, rank, e[0], e[1], e[2], e[3], int(there.sum())))


def _train_overlapped(args, data_loader, model, log, rank, world, t_obs, t_pred, losses) -> List[float]:
    """The same schedule with the loader off the critical path (desire_amd/prefetch.py): a background thread runs next_batch into pinned
    staging and uploads on a copy stream `--prefetch` batches ahead; the step's loss is read back ONE STEP LATE, so the host enqueues
    step i + 1 while the device still runs step i.  Same batches, same seeds, same optimiser steps as the serial loop -- the log line of
    step i is printed after step i + 1 has been enqueued, nothing else differs (tests/test_gpu_prefetch.py compares the two)."""
    from .model import dims_from_args
    from .prefetch import WindowFeeder
    mno = dims_from_args(model.args, 1, True).mno
    feeder = WindowFeeder(data_loader, t_obs, t_pred, device=model.device, depth=int(args.prefetch), num_epochs=args.num_epochs,
                          shard=(rank, world), mno=mno, max_batches=int(args.max_steps or 0))
    nb = data_loader.num_batches
    pending = None                                        # (PendingLoss, epoch, batch, start time) of the previous step

    def settle(p):
        terms, epoch, batch, start = p[0].get(), p[1], p[2], p[3]
        losses.append(terms["loss"])
        if rank == 0:
            log("{}/{} (epoch {}), train_loss = {:.3f}, time/batch = {:.3f}".format(
                epoch * nb + batch, args.num_epochs * nb, epoch, terms["loss"], time.time() - start))
            sys.stdout.flush()

    steps = 0
    last = None
    try:
        for bt in feeder:
            start = time.time()
            model.learning_rate = lr_at_epoch(args, bt.epoch)
            if last is not None and bt.epoch != last[2] and getattr(args, "report_ade", False):
                _report_ade(args, model, last[0], last[1], last[2], rank, log)
            bt.wait()
            pl = model.train_step_device(bt.past, bt.fut, seed=args.seed + steps * world + rank, sync=False)
            if getattr(args, "report_ade", False):        # the epoch's last batch is evaluated after the feeder has moved on: keep a copy
                last = (bt.past.clone(), bt.fut.clone(), bt.epoch)
            bt.release()
            if pending is not None:
                settle(pending)
            pending = (pl, bt.epoch, bt.index, start)
            steps += 1
            if rank == 0 and should_save(bt.epoch, bt.index, nb, args.save_every):
                path = os.path.join(args.save_dir, "social_model-%d.npz" % (bt.epoch * nb + bt.index))
                model.save(path)
                log("model saved to {}".format(path))
    finally:
        feeder.close()
    if pending is not None:
        settle(pending)
    if last is not None and getattr(args, "report_ade", False):
        _report_ade(args, model, last[0], last[1], last[2], rank, log)
    return losses


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.skip_padding and args.keep_padding:
        raise SystemExit("--skip_padding and --keep_padding exclude each other")
    # (dims.flags are derived in desire_amd/model.py: _flags_from_args -- padding skipped unless --keep_padding)
    import torch
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # DESIRE_DIST_BACKEND=gloo + DESIRE_ONE_GPU=1: the multi-rank path on a box with a single GPU (all ranks share cuda:0 and the
        # flat-gradient all-reduce runs over gloo, because RCCL refuses two ranks on one device); a test switch, never a deployment
        backend = os.environ.get("DESIRE_DIST_BACKEND", "nccl")
        local = 0 if os.environ.get("DESIRE_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    try:
        from .model import DESIREModel
        model = DESIREModel(args, seed=args.seed)
        train(args, model=model)
        if dist.is_initialized():
            # data-parallel invariant: every rank applied the same averaged gradient, so every rank holds the same weights
            w = model.sync_weights()
            chk = float(sum(float(np.abs(np.asarray(v, np.float64)).sum()) for v in w.values()))
            print("rank {} weights_checksum = {:.9e}".format(dist.get_rank(), chk))
            sys.stdout.flush()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
