"""Entry points `train.py --config` / `train_iterable.py --config`: host logic on CPU, and an
end-to-end run on the GPU with synthetic wav files (workspace tree, checkpoint keys, console
lines and artefacts of the reference, train.py:94-109,136-151,198-250,254-307)."""
import configparser
import json
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from conftest import GOLDEN, REPO  # noqa: E402


def _sine_wav(path, seconds, sr, f0, stereo=False):
    from scipy.io import wavfile
    t = np.arange(int(seconds * sr)) / sr
    a = (0.5 * np.sin(2 * np.pi * f0 * t) + 0.2 * np.sin(2 * np.pi * 2.7 * f0 * t)).astype(np.float32)
    pcm = (a * 32767).astype(np.int16)
    if stereo:
        pcm = np.stack([pcm, -pcm], axis=1)
    wavfile.write(str(path), sr, pcm)
    return a


def _dataset(tmp_path, sr=8000):
    tmp_path.mkdir(parents=True, exist_ok=True)
    (tmp_path / "audio").mkdir()
    (tmp_path / "test_audio").mkdir()
    _sine_wav(tmp_path / "audio" / "a.wav", 1.3, sr, 220.0)
    _sine_wav(tmp_path / "audio" / "b.wav", 0.9, sr, 330.0, stereo=True)
    _sine_wav(tmp_path / "test_audio" / "t.wav", 0.5, sr, 440.0)
    return tmp_path


def _ini(tmp_path, iterable=False, **over):
    cfg = configparser.ConfigParser(allow_no_value=True)
    cfg.read(os.path.join(REPO, "default_iterable.ini" if iterable else "default.ini"))
    cfg["audio"].update(sampling_rate="8000", hop_length="64", segment_length="256")
    cfg["dataset"]["datapath"] = str(tmp_path)
    cfg["VAE"].update(latent_dim="8", n_units="128")
    cfg["training"].update(batch_size="64", checkpoint_interval="2", learning_rate="0.001")
    if iterable:
        cfg["training"]["total_num_frames"] = str(64 * 7)
        cfg["training"]["checkpoint_interval"] = "3"
    else:
        cfg["training"].update(epochs="4", save_best_model_after="1")
    cfg["extra"]["description"] = "unit"
    cfg["mi355x"].update(tensorboard="False", loss_ring="4")
    for k, v in over.items():
        sec, key = k.split("__")
        cfg[sec][key] = v
    p = tmp_path / "run.ini"
    with open(p, "w") as f:
        cfg.write(f)
    return p


def test_streaming_resampler_is_the_windowed_sinc_interpolation_it_restates():
    """`data._resample_sinc_hann` restates `torchaudio.functional.resample`'s published algorithm (dataset.py:50-51 calls it
    with the defaults: sinc_interp_hann, width 6, rolloff 0.99).  torchaudio is absent, so this does NOT pin it to the
    reference's output; it holds the filter-bank / strided-convolution form to the interpolation formula it implements,
    evaluated directly in float64 at sampled output positions, for up- and down-sampling rate pairs, plus the output
    length rule and a tone below both Nyquist rates."""
    import math
    from rawaudiovae_kelsey_amd import data as D

    def direct(a, sr_in, sr_out, pos, lpw=6, rolloff=0.99):
        g = math.gcd(sr_in, sr_out)
        orig, new = sr_in // g, sr_out // g
        base = min(orig, new) * rolloff
        out = []
        for m in pos:
            t = np.clip((np.arange(len(a)) - m * orig / new) / orig * base, -lpw, lpw)
            w = np.cos(t * math.pi / lpw / 2) ** 2
            tt = t * math.pi
            s = np.where(tt == 0, 1.0, np.sin(tt) / np.where(tt == 0, 1.0, tt))
            out.append(float(np.sum(a * s * w) * base / orig))
        return np.array(out)
    rng = np.random.default_rng(0)
    for si, so in [(48000, 44100), (44100, 22050), (22050, 44100), (32000, 44100), (44100, 16000)]:
        a = rng.standard_normal(3000).astype(np.float32)
        y = D._resample_sinc_hann(a, si, so)
        g = math.gcd(si, so)
        assert y.dtype == np.float32 and len(y) == math.ceil((so // g) * len(a) / (si // g))
        pos = [0, 1, 2, 57, 500, len(y) // 2, len(y) - 3, len(y) - 1]
        assert np.abs(y[pos] - direct(a.astype(np.float64), si, so, pos)).max() < 5e-6
    tone = np.sin(2 * np.pi * 1000.0 * np.arange(48000) / 48000).astype(np.float32)
    y = D._resample_sinc_hann(tone, 48000, 44100)
    exp = np.sin(2 * np.pi * 1000.0 * np.arange(len(y)) / 44100)
    assert np.abs(y[200:-200] - exp[200:-200]).max() < 2e-3
    assert D._resample_sinc_hann(tone, 44100, 44100) is tone      # same rate: untouched
    # a long signal goes through the chunked convolution with the same result as one piece
    long_a = rng.standard_normal(200000).astype(np.float32)
    assert np.array_equal(D._resample_sinc_hann(long_a, 48000, 44100, chunk=257), D._resample_sinc_hann(long_a, 48000, 44100))


def test_wav_io_and_frame_count_host_logic(tmp_path):
    from rawaudiovae_kelsey_amd import data as D
    a = _sine_wav(tmp_path / "m.wav", 0.25, 8000, 200.0)
    got = D.load_audio_mono(tmp_path / "m.wav", 8000)
    assert got.dtype == np.float32 and len(got) == len(a) and np.abs(got - a).max() < 1e-4
    _sine_wav(tmp_path / "s.wav", 0.25, 8000, 200.0, stereo=True)
    assert np.abs(D.load_audio_mono(tmp_path / "s.wav", 8000)).max() < 1e-4      # L + (-L) averages to 0
    assert np.abs(D.load_audio_ch0(tmp_path / "s.wav", 8000) - a).max() < 1e-4   # streaming path: channel 0
    assert len(D.load_audio_mono(tmp_path / "m.wav", 4000)) == len(a) // 2        # resampled
    D.write_wav(tmp_path / "o.wav", a, 8000)
    assert np.array_equal(D.read_wav(tmp_path / "o.wav")[0], a)
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        d = json.load(f)["dataset"]
    assert D.frame_count(d["n_samples"], d["segment_length"], d["hop"]) == (d["len"], d["padded"])
    with pytest.raises(ValueError):
        D.frame_count(d["n_samples"], d["bad_segment_length"], d["hop"])


def test_workspace_numbering_and_missing_paths(tmp_path):
    sys.path.insert(0, REPO)
    import train as T
    w0 = T.make_workspace(tmp_path, "desc", 0)
    w1 = T.make_workspace(tmp_path, "desc", 0)
    assert w0.name == "run-000" and w1.name == "run-001" and w1.parent.name == "desc"
    ini = _ini(_dataset(tmp_path / "d1"), dataset__datapath=str(tmp_path / "nope"))
    with pytest.raises(FileNotFoundError):
        T.main(["--config", str(ini)])
    with pytest.raises(SystemExit):
        T.main(["--config", str(tmp_path / "missing.ini")])


@pytest.mark.gpu
def test_train_py_end_to_end(tmp_path, capsys):
    sys.path.insert(0, REPO)
    import train as T
    from rawaudiovae_kelsey_amd import data as D
    ds = _dataset(tmp_path)
    workdir = T.main(["--config", str(_ini(ds))])
    out = capsys.readouterr().out
    assert workdir.name == "run-000" and workdir.parent.name == "unit"
    for rel in ("config.ini", "model/checkpoints/ckpt_00002", "model/checkpoints/ckpt_00004", "model/last_model.pt",
                "model/best_model.pt", "audio_logs/test_audio.txt", "audio_logs/test_original.wav",
                "audio_logs/test_reconst_00002.wav", "audio_logs/test_reconst_00004.wav", "logs"):
        assert (workdir / rel).exists(), rel
    assert "Epoch 0/3" in out and "----------" in out and "Checkpoint - Epoch 2" in out
    assert "Total number of audio frames:" in out and "Training Finished: Saved the last model" in out
    losses = [float(l.split("Total loss: ")[1].split(" - ")[0]) for l in out.splitlines() if l.startswith("====> Epoch")]
    assert len(losses) == 4 and losses[-1] < losses[0]
    ck = torch.load(workdir / "model/checkpoints/ckpt_00004", weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 3
    assert list(ck["state_dict"]) == ["fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight",
                                      "fc22.bias", "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias"]
    n_frames, _ = D.frame_count(int(1.3 * 8000) + int(0.9 * 8000), 256, 64)
    steps = 4 * ((n_frames + 63) // 64)
    assert int(ck["optimizer"]["state"][0]["step"]) == steps
    opt = torch.optim.Adam([torch.nn.Parameter(v.clone()) for v in ck["state_dict"].values()], lr=1e-3)
    opt.load_state_dict(ck["optimizer"])          # the optimizer dict is in torch.optim.Adam's own format
    m = torch.load(workdir / "model/last_model.pt", weights_only=False)
    assert type(m).__module__ == "rawvae.model"
    rec, sr = D.read_wav(workdir / "audio_logs/test_reconst_00004.wav")
    orig, _ = D.read_wav(workdir / "audio_logs/test_original.wav")
    assert sr == 8000 and len(rec) == ((len(orig) + 255) // 256) * 256 and np.isfinite(rec).all()
    cfg = configparser.ConfigParser(allow_no_value=True)
    cfg.read(workdir / "config.ini")
    assert cfg["dataset"]["workspace"] == str(workdir.resolve()) and cfg["VAE"]["device_name"]
    assert cfg["dataset"]["total_frames"] == str((int(1.3 * 8000) + int(0.9 * 8000)) // 256)


@pytest.mark.gpu
def test_train_iterable_end_to_end(tmp_path, capsys):
    sys.path.insert(0, REPO)
    import train_iterable as TI
    ds = _dataset(tmp_path)
    workdir = TI.main(["--config", str(_ini(ds, iterable=True))])
    out = capsys.readouterr().out
    batch_lines = [l for l in out.splitlines() if l.startswith("====> Batch:")]
    assert len(batch_lines) == 7 and batch_lines[0].startswith("====> Batch: 0 - Loss: ")
    for rel in ("console_log", "model/checkpoints/ckpt_00003", "model/checkpoints/ckpt_00006",
                "model/checkpoints/ckpt_00007", "model/last_model.pt", "audio_logs/test_reconst_00003.wav"):
        assert (workdir / rel).exists(), rel
    ck = torch.load(workdir / "model/checkpoints/ckpt_00006", weights_only=False)
    assert set(ck) == {"batch_id", "state_dict", "optimizer"} and ck["batch_id"] == 6
    assert "====> Batch: 6" in open(workdir / "console_log").read()


@pytest.mark.gpu
def test_train_iterable_fp8_weight_path_and_fp16_slabs(tmp_path, capsys):
    """BASELINE configs[4]: the streaming entry point with the fp8 forward weight path ([mi355x] fp8 = True), here
    together with fp16 split-K slabs: the run completes with the same artefacts and a loss that descends like the
    bf16 run's (same data order, same seeds)."""
    sys.path.insert(0, REPO)
    import train_iterable as TI
    ds = _dataset(tmp_path)

    def run(**kw):
        TI.main(["--config", str(_ini(ds, iterable=True, **kw))])
        out = capsys.readouterr().out
        return [float(l.split("Loss: ")[1].split()[0]) for l in out.splitlines() if l.startswith("====> Batch:")]
    ref = run()
    got = run(mi355x__fp8="True", mi355x__wgrad_slabs="fp16")
    assert len(got) == len(ref) == 7 and all(np.isfinite(got))
    np.testing.assert_allclose(got, ref, rtol=2e-2)
    with pytest.raises(ValueError, match="wgrad_slabs"):
        TI.main(["--config", str(_ini(ds, iterable=True, mi355x__wgrad_slabs="int8"))])


@pytest.mark.gpu
def test_streaming_frames_order_and_file_boundaries(tmp_path):
    """Unshuffled stream == hop frames of file a (ch0), then file b, cycled (dataset.py:53-84)."""
    from oracle import vae_oracle as O
    from rawaudiovae_kelsey_amd import data as D
    ds = _dataset(tmp_path)
    files = sorted((ds / "audio").glob("*.wav"))
    ref = np.concatenate([O.hop_frames(D.load_audio_ch0(f, 8000), 256, 64) for f in files] * 2)
    st = D.StreamingFrames(files, 8000, 64, 256, "cuda", shuffle=False)
    got = torch.cat(list(st.batches(50, 9))).cpu().numpy()
    np.testing.assert_array_equal(got, ref[:450])


@pytest.mark.gpu
def test_device_audio_epoch_is_a_permutation_with_ragged_tail():
    from oracle import vae_oracle as O
    from rawaudiovae_kelsey_amd import data as D
    rng = np.random.default_rng(2)
    audio = rng.uniform(-1, 1, 64 * 37 + 11).astype(np.float32)
    d = D.DeviceAudio(audio, 256, 64)
    ref = O.hop_frames(audio, 256, 64)
    assert len(d) == len(ref)
    bs = [b.cpu().numpy() for b in d.batches(16, shuffle=True, generator=torch.Generator().manual_seed(1))]
    assert [len(b) for b in bs] == [16] * (len(ref) // 16) + [len(ref) % 16]
    got = np.concatenate(bs)
    key = lambda a: sorted(map(bytes, a))
    assert key(got) == key(ref)       # every frame exactly once
    assert not np.array_equal(got, ref)  # and shuffled
    with pytest.raises(ValueError):
        D.DeviceAudio(audio, 250, 64)


@pytest.mark.gpu
def test_train_py_data_parallel_two_ranks(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 train.py`: both ranks on one GPU with gloo carrying the
    all-reduces (on a multi-GPU node the same code runs over RCCL).  Rank 0 alone writes the workspace; the
    replicas must stay identical (train.py raises otherwise); per-rank batch 32 => global batch 64."""
    import subprocess
    ds = _dataset(tmp_path)
    ini = _ini(ds, training__batch_size="32")
    env = dict(os.environ, RV_DIST_BACKEND="gloo")
    import socket
    with socket.socket() as so:      # a port nobody listens on right now
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(REPO, "train.py"), "--config", str(ini)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = r.stdout
    assert out.count("Training Finished: Saved the last model") == 1 and "(data parallel)" in out
    runs = sorted((ds / "unit").iterdir())
    assert [p.name for p in runs] == ["run-000"]                       # one workspace, made by rank 0
    for rel in ("config.ini", "model/checkpoints/ckpt_00004", "model/last_model.pt", "audio_logs/test_reconst_00004.wav"):
        assert (runs[0] / rel).exists(), rel
    losses = [float(l.split("Total loss: ")[1].split(" - ")[0]) for l in out.splitlines() if l.startswith("====> Epoch")]
    assert len(losses) == 4 and losses[-1] < losses[0]
    ck = torch.load(runs[0] / "model/checkpoints/ckpt_00004", weights_only=False)
    from rawaudiovae_kelsey_amd import data as D
    n_frames, _ = D.frame_count(int(1.3 * 8000) + int(0.9 * 8000), 256, 64)
    per_epoch = n_frames // 64 + (1 if (n_frames % 64) // 2 else 0)    # full global batches + the evenly split tail
    assert int(ck["optimizer"]["state"][0]["step"]) == 4 * per_epoch


def test_sharded_batches_cover_the_epoch_once():
    """Index logic of the data-parallel epoch (no GPU): every frame of the permutation is used at most once,
    all ranks get the same batch sizes, and at most world - 1 frames are left out."""
    from rawaudiovae_kelsey_amd.data import DeviceAudio

    class FakeAudio(DeviceAudio):
        def __init__(self, n):
            self.n_frames, self.device = n, torch.device("cpu")

        def gather(self, index, out=None, stream=None):
            return index

    for n, bs, world in ((1000, 64, 4), (257, 32, 8), (64, 64, 2), (10, 4, 3)):
        seen, sizes = [], []
        for rank in range(world):
            g = torch.Generator().manual_seed(7)
            got = list(FakeAudio(n).sharded_batches(bs, rank, world, generator=g))
            sizes.append([len(b) for b in got])
            seen += [int(v) for b in got for v in b]
        assert all(s == sizes[0] for s in sizes)
        assert len(seen) == len(set(seen)) and n - len(seen) < world


def test_kelsey_iterable_ini_parses_like_the_reference():
    """The reference's only working streaming config (kelsey_iterable.ini: free-text [notes] lines are keys
    without values, train_iterable.py:40 allow_no_value=True) goes through this build's parser with the run
    length the reference derives from it (train_iterable.py:70-74)."""
    import train as T
    from conftest import REPO
    cfg = T.read_config(os.path.join(REPO, "kelsey_iterable.ini"))
    assert cfg["training"].getint("batch_size") == 4096
    assert cfg["training"].getint("checkpoint_interval") == 754
    total = int(cfg["training"].getint("total_num_frames") / cfg["training"].getint("batch_size"))
    assert total == 37674 and total // 754 == 49   # int(154314100 / 4096); the notes in the reference say 37676
    notes = dict(cfg["notes"])
    assert notes["additional_notes"] == "" and sum(v is None for v in notes.values()) >= 2   # lines without "=" or ":" are valueless keys
    assert cfg["VAE"].getint("latent_dim") == 256 and cfg["audio"].getint("hop_length") == 128
