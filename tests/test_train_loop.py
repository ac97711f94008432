"""Host-side training loop (desire_amd/train.py) against the reference's train.py:24-207 behaviour."""
import os

import numpy as np
import pytest

from desire_amd import train as T


def test_parser_defaults_are_the_reference_flags():
    a = T.build_parser().parse_args([])
    # train.py:30-85
    ref = dict(rnn_size=512, num_layers=1, model="gru", batch_size=10, seq_length=8, num_epochs=100, save_every=400,
               grad_clip=10.0, learning_rate=0.005, decay_rate=0.95, keep_prob=0.8, embedding_size=64, neighborhood_size=32,
               grid_size=4, max_num_obj=60, leave_dataset=5, latent_size=128, e_dim=256, d_dim=16, stride=1)
    for k, v in ref.items():
        assert getattr(a, k) == v, k


def test_lr_decay_and_save_cadence():
    a = T.build_parser().parse_args([])
    assert T.lr_at_epoch(a, 0) == 0.005
    assert abs(T.lr_at_epoch(a, 3) - 0.005 * 0.95 ** 3) < 1e-15              # train.py:122-126
    nb = 58
    saves = [(e, b) for e in range(20) for b in range(nb) if T.should_save(e, b, nb, 400)]
    assert saves[0] == (6, 52) and all((e * nb + b) % 400 == 0 for e, b in saves)   # train.py:197-199, never at step 0
    assert not T.should_save(0, 0, nb, 400)


def test_split_windows():
    x = [np.arange(5 * 3 * 3, dtype=np.float64).reshape(5, 3, 3)]
    p, f = T.split_windows(x, 2)
    assert p[0].shape == (2, 3, 3) and f[0].shape == (3, 3, 3)
    np.testing.assert_array_equal(np.concatenate([p[0], f[0]]), x[0])


def _synthetic_video(n_frames, mno, n_ids, rng):
    """(frames, MNO, 3) array in the loader's preprocessed layout: smooth tracks, ids 1..n_ids in the first slots."""
    t = np.arange(n_frames)[:, None]
    x0, y0 = rng.uniform(300, 1100, n_ids), rng.uniform(300, 900, n_ids)
    vx, vy = rng.normal(0, 3, n_ids), rng.normal(0, 3, n_ids)
    arr = np.zeros((n_frames, mno, 3))
    arr[:, :n_ids, 0] = np.arange(1, n_ids + 1)
    arr[:, :n_ids, 1] = x0 + vx * t
    arr[:, :n_ids, 2] = y0 + vy * t
    return arr


@pytest.mark.gpu
@pytest.mark.parametrize("operands", ["", "x3", "keep", "x3+keep"])      # fp32; split-bf16 operands (dims.bf16 = 2); padding skipped (the default: DESIRE_FLAG_COMPACT_*) / --keep_padding
def test_training_loop_runs_saves_and_learns(tmp_path, operands):
    from desire_amd.data_loader import DataLoader
    from desire_amd.model import DESIREModel
    rng = np.random.default_rng(0)
    frames = [_synthetic_video(120, 8, 6, rng), _synthetic_video(90, 8, 5, rng)]
    a = T.build_parser().parse_args(["--batch_size", "4", "--seq_length", "4", "--pred_length", "6", "--max_num_obj", "8",
                                     "--d_dim", "64", "--latent_size", "64", "--num_samples", "3", "--num_epochs", "3",
                                     "--save_every", "5", "--learning_rate", "0.0005", "--neighborhood_size", "256",
                                     "--save_dir", str(tmp_path / "save")] + (["--bf16", "x3"] if "x3" in operands else []) +
                                    (["--keep_padding"] if "keep" in operands else []))
    dl = DataLoader(a.batch_size, a.seq_length + a.pred_length, a.max_num_obj, frames=frames)
    assert dl.num_batches > 0
    import random
    random.seed(0)
    lines = []
    losses = T.train(a, data_loader=dl, log=lines.append)
    assert len(losses) == a.num_epochs * dl.num_batches and np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < np.mean(losses[:3])
    assert os.path.exists(tmp_path / "save" / "config.pkl")
    saved = sorted((f for f in os.listdir(tmp_path / "save") if f.endswith(".npz")), key=lambda f: int(f.split("-")[1][:-4]))
    assert saved and saved[0] == "social_model-5.npz"
    assert any("train_loss" in l for l in lines) and any("model saved" in l for l in lines)
    m2 = DESIREModel.restore(a, str(tmp_path / "save" / saved[-1]))
    x, _, _ = dl.next_batch(random_update=False)
    past, fut = T.split_windows(x, a.seq_length)
    Y, s = m2.forward(past, fut, seed=0)
    assert bool(np.isfinite(Y.cpu().numpy()).all())


@pytest.mark.gpu
def test_training_with_the_reference_default_slot_count(tmp_path):
    """train.py's default --max_num_obj is 60: the model pads it to 64 slots, i.e. one 64-row IOC tile per (scene, sample)
    in the forward AND the backward pass (crowded SDD scenes need it: up to 66 objects per frame)."""
    from desire_amd.data_loader import DataLoader
    rng = np.random.default_rng(1)
    frames = [_synthetic_video(60, 60, 45, rng)]
    a = T.build_parser().parse_args(["--batch_size", "2", "--seq_length", "4", "--pred_length", "5", "--d_dim", "64",
                                     "--latent_size", "64", "--num_samples", "2", "--num_epochs", "2", "--learning_rate", "0.0005",
                                     "--neighborhood_size", "256", "--save_dir", str(tmp_path / "save")])
    assert a.max_num_obj == 60
    dl = DataLoader(a.batch_size, a.seq_length + a.pred_length, a.max_num_obj, frames=frames)
    import random
    random.seed(0)
    losses = T.train(a, data_loader=dl, log=lambda l: None)
    assert len(losses) == a.num_epochs * dl.num_batches and np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < np.mean(losses[:3])


@pytest.mark.gpu
def test_training_on_a_real_sdd_slice_improves_best_of_k_ade(tmp_path):
    """End to end on real Stanford Drone Dataset frames (bookstore/video6, the 160 preprocessed frames kept as a loader
    golden): reference-layout DataLoader -> desire_amd.train loop -> prior sampling -> ADE/FDE harness."""
    import random
    from desire_amd.data_loader import DataLoader
    from desire_amd.model import DESIREModel
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loader_bookstore6_T8.npz"))
    frames = [z["data0"]]                                            # [160, 32, 3] = [id, x_px, y_px], zero rows = absent
    a = T.build_parser().parse_args(["--batch_size", "4", "--seq_length", "8", "--pred_length", "12", "--max_num_obj", "32",
                                     "--d_dim", "64", "--latent_size", "64", "--num_samples", "5", "--num_epochs", "12",
                                     "--save_every", "1000", "--learning_rate", "0.001", "--neighborhood_size", "200",
                                     "--save_dir", str(tmp_path / "save")])
    a.img_width, a.img_height = 1424.0, 1088.0                       # bookstore frame size (pixels -> normalised units)
    dl = DataLoader(a.batch_size, a.seq_length + a.pred_length, a.max_num_obj, frames=frames)
    random.seed(1)
    xval, _, _ = dl.next_batch(random_update=False)
    past, fut = T.split_windows(xval, a.seq_length)
    model = DESIREModel(a, seed=5)

    pw = np.stack(past)                                              # [n, T_obs, 32, 3]
    fw = np.stack(fut)
    there = ((pw[:, -1, :, 0] != 0) & (fw[:, :, :, 0] != 0).all(1)).reshape(-1)   # tracked at the last observed frame and all future frames
    assert there.sum() >= 8

    def ade():
        Y, _ = model.forward(past, None, seed=7)                     # prior sampling: no future given
        return model.evaluate(Y, fut)[there].mean(0)                 # (ADE mean-of-K, FDE mean-of-K, ADE best-of-K, FDE best-of-K)

    before = ade()
    dl.reset_batch_pointer()
    losses = T.train(a, data_loader=dl, model=model, log=lambda s: None)
    model.sync_weights()
    after = ade()
    print("loss %.3f -> %.3f; [ADE mean-of-K, FDE mean-of-K, ADE best-of-K, FDE best-of-K] (normalised units) before %s after %s"
          % (losses[0], losses[-1], np.round(before, 4), np.round(after, 4)))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
    assert after[0] < before[0] and after[2] < before[2]             # mean-of-K and best-of-K ADE both improve


@pytest.mark.gpu
def test_command_line_entry_on_a_csv_directory(tmp_path, golden_dir):
    """`python -m desire_amd.train --data_dir ...` end to end: the reference's directory layout (data/<scene>/<video>/
    annotations_processed.csv), preprocessing, the loop, the log lines, a checkpoint and config.pkl."""
    import subprocess
    import sys
    g = np.load(os.path.join(golden_dir, "loader_bookstore6_T8.npz"))
    vid = tmp_path / "data" / "bookstore" / "video6"
    vid.mkdir(parents=True)
    np.savetxt(vid / "annotations_processed.csv", g["csv"].astype(np.float64), delimiter=",", fmt="%.1f")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "desire_amd.train", "--data_dir", str(tmp_path / "data") + "/", "--save_dir", str(tmp_path / "save"),
           "--batch_size", "4", "--seq_length", "8", "--pred_length", "12", "--max_num_obj", "32", "--d_dim", "64", "--latent_size", "64",
           "--num_samples", "3", "--num_epochs", "3", "--save_every", "4", "--learning_rate", "0.0005",
           "--neighborhood_size", "64"]
    p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count("train_loss") == 6 and "model saved to" in p.stdout      # 160 frames -> 2 batches per epoch (:175-184)
    assert os.path.exists(tmp_path / "save" / "config.pkl") and os.path.exists(tmp_path / "save" / "social_model-4.npz")
    assert os.path.exists(tmp_path / "data" / "trajectories.cpkl")       # the reference's preprocessed pickle, written on first use


@pytest.mark.gpu
def test_config4_shape_two_ranks_mixed_sdd_scenes(tmp_path, golden_dir):
    """BASELINE configs[4] at its stated shape: "full SDD mixed-scene training loop (fwd+bwd), 512 agents/step,
    K=20, IOC refinement on" -- `python -m desire_amd.train` over the eight-scene mix (one slice of every SDD scene, the CSVs the
    reference loader read for tests/golden/loader_mixed8_T20.npz), --num_samples 20 --d_dim 128, batch 8 windows x 64 slots (the
    reference's --max_num_obj 40 padded to the 64-row tile) = 512 agent slots per step, data-parallel over TWO ranks (4 windows each,
    gradients averaged by the flat all-reduce; gloo, because both ranks share the one GPU of the test box): the loss goes down,
    both ranks end with the same weights, ADE / FDE are reported."""
    import re
    import subprocess
    import sys
    g = np.load(os.path.join(golden_dir, "loader_mixed8_T20.npz"))
    for i, name in enumerate(g["order"]):
        scene, video = str(name).split("/") if "/" in str(name) else (str(name), "video0")
        vid = tmp_path / "data" / scene / video
        vid.mkdir(parents=True)
        np.savetxt(vid / "annotations_processed.csv", g["csv%d" % i].astype(np.float64), delimiter=",", fmt="%.1f")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DESIRE_DIST_BACKEND="gloo", DESIRE_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", "-m", "desire_amd.train", "--data_dir", str(tmp_path / "data") + "/", "--save_dir", str(tmp_path / "save"),
           "--batch_size", "8", "--seq_length", "8", "--pred_length", "12", "--max_num_obj", "40", "--d_dim", "128", "--latent_size", "128",
           "--num_samples", "20", "--num_epochs", "5", "--learning_rate", "0.0005", "--neighborhood_size", "160", "--leave_dataset", "99",
           "--report_ade"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    losses = [float(m) for m in re.findall(r"train_loss = ([-0-9.eE+]+)", p.stdout)]
    assert len(losses) >= 10 and np.isfinite(losses).all(), p.stdout[-2000:]
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    chk = dict(re.findall(r"rank (\d) weights_checksum = ([-0-9.eE+]+)", p.stdout))
    assert set(chk) == {"0", "1"} and chk["0"] == chk["1"], chk            # same averaged gradient on both ranks, every step
    ade = re.findall(r"ADE/FDE mean-of-K = ([0-9.]+) / ([0-9.]+), best-of-K = ([0-9.]+) / ([0-9.]+)", p.stdout)
    assert len(ade) >= 5 and all(float(b[2]) <= float(b[0]) + 1e-9 for b in ade)      # best-of-K never worse than mean-of-K
