#!/usr/bin/env python3
"""`python train.py --config default.ini` -- the reference's entry point
(/root/reference/train.py) on the MI355X path.

Same CLI, same ini sections/keys (`default.ini`), same workspace tree
(`<datapath>/<description>/run-NNN/{config.ini, model/checkpoints/ckpt_%05d,
model/best_model.pt|last_model.pt, logs/, audio_logs/{<test_dataset>.txt,
test_original.wav, test_reconst_%05d.wav}}`), same checkpoint dict keys
(`epoch`, `state_dict`, `optimizer`) and the same console lines.  What differs:

  * the inner loop (train.py:179-196) is `TrainEngine.step` -- one host call per batch that
    runs forward, fused loss, backward and Adam in hand-written gfx950 kernels; the waveform
    lives in HBM and batches are gathered on the device (`DeviceAudio`);
  * per-batch losses are read back from a device ring once per `ring` batches instead of a
    `loss.item()` sync per batch;
  * latent bugs of the reference are not reproduced: the device check uses
    `device.type`, `generate_test` is parsed as a boolean, test predictions are concatenated as
    a list, and "best model" means the lowest epoch loss seen so far;
  * librosa / soundfile are replaced by scipy wav I/O; TensorBoard is used when importable.

An optional `[mi355x]` section adds `seed`, `loss_ring`, `tensorboard`, `fp8` (fc1 / fc4 forward on e4m3 operands,
BASELINE configs[4]) and `wgrad_slabs` (`fp32` | `fp16` split-K partial sums) keys.

Data-parallel training (no counterpart in the reference, which is single-process): launch with
`python -m torch.distributed.run --nproc-per-node N train.py --config default.ini`, one process per
GPU.  `batch_size` is then the per-GPU batch (global batch N * batch_size); every rank keeps the
waveform in its own HBM and takes its slice of each global batch of the shared epoch permutation;
gradients are averaged over ranks inside `engine.step_ddp` (RCCL); rank 0 alone writes the
workspace.  The reported epoch loss is the sum over global batches of the rank-mean batch loss.
"""
import argparse
import configparser
import contextlib
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from rawvae.model import VAE  # noqa: E402
from rawaudiovae_kelsey_amd import data as D  # noqa: E402
from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402


def read_config(path):
    config = configparser.ConfigParser(allow_no_value=True)
    if not config.read(path):
        print('Config File Not Found at {}'.format(path))
        sys.exit(1)
    return config


def engine_options(hw):
    """TrainEngine keyword arguments from the optional [mi355x] section: `fp8 = True` runs the fc1 / fc4 forward
    GEMMs and fc4's backward on fp8 (e4m3) operands (BASELINE configs[4]; `fp8 = fwd`: the forward only); `wgrad_slabs = fp16` (default) stores the split-K partial sums
    of the two large weight gradients as block-floating-point fp16 (one power-of-two scale per wave tile), `fp32`
    keeps them fp32."""
    kw = {}
    fp8 = str(hw.get('fp8', 'False')).lower()
    if fp8 in ('1', 'true', 'yes', 'full'):
        kw['fp8'] = True
    elif fp8 == 'fwd':
        kw['fp8'] = 'fwd'    # the two forward GEMMs only (round 3's path)
    slabs = str(hw.get('wgrad_slabs', 'fp16')).lower()
    if slabs not in ('fp32', 'fp16'):
        raise ValueError("[mi355x] wgrad_slabs = {} (expected fp32 or fp16)".format(slabs))
    kw['slab_dtype'] = slabs
    return kw


def require_gpu(local_rank=0):
    if not torch.cuda.is_available():
        raise RuntimeError("train.py (MI355X build) needs a GPU: torch.cuda.is_available() is False")
    index = local_rank % torch.cuda.device_count()   # the modulo only matters for one-GPU rehearsals
    torch.cuda.set_device(index)
    return torch.device('cuda', index)


class DataParallel:
    """Rank bookkeeping + the per-step exchange for train.py under torch.distributed.run."""

    def __init__(self, device):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        self.active = self.world > 1
        self.device = device
        self.comm = None
        self.engines = []      # every engine the exchange is attached to (the full-batch one and the ragged tail's)
        if self.active:
            backend = os.environ.get("RV_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
            if backend == "nccl" and os.environ.get("RV_DDP", "native") != "torch":
                # The library-driven RCCL step (one host call per batch).  All ranks use it or none does: a
                # rank whose communicator or self-test fails must not leave the others inside a collective,
                # so the outcome is agreed on over the torch.distributed group (as bench.py does) and a
                # failure anywhere moves every rank to the torch.distributed exchange (ddp.ddp_step).
                from rawaudiovae_kelsey_amd import ddp
                ok, why = 1, ""
                try:
                    self.comm = ddp.RcclComm()
                    self.comm.self_test(device)
                except Exception as exc:
                    ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if why or self.rank == 0:
                        print("rank %d: library-driven RCCL step unavailable%s; all ranks use the torch.distributed "
                              "exchange" % (self.rank, " (" + why + ")" if why else " on another rank"), flush=True)
                    if self.comm is not None and ok:
                        self.comm.destroy()
                    self.comm = None

    @property
    def main(self):
        return self.rank == 0

    def share(self, obj):
        """Rank 0's object on every rank."""
        if not self.active:
            return obj
        box = [obj if self.main else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def rendezvous(self):
        """Host-side meeting point of all ranks behind work that rank 0 does alone (the evaluation pass and the checkpoint
        files of a checkpoint epoch).  Without it the other ranks go straight into the next epoch's first data-parallel
        step and spin on its cross-stream flags behind rank 0 -- whose collectives are not enqueued while it writes files --
        and a checkpoint that takes longer than the flag waits' bound (RV_DDP_WAIT_MS, thirty seconds by default: a large
        test set, a slow or shared file system) would poison every plan of a healthy run.  Waiting HERE costs nothing: the
        ranks block in a host call with no step in flight and no bound."""
        if self.active:
            self.dist.barrier()

    def check(self):
        """Health of the library-driven step, before results are read back (every rank calls it at the same points).
        A flag wait that ran out on ANY engine of ANY rank (engine.ddp_timeouts: a peer more than RV_DDP_WAIT_MS behind,
        or gone) has poisoned that plan -- its updates are withheld on the device -- while the other ranks went on.
        Nothing after that can be trusted and a rank that raised alone would leave its peers inside the next collective,
        so the ranks agree on the worst count and ALL leave, non-zero."""
        if not self.active or self.comm is None:
            return
        n = sum(e.ddp_timeouts() for e in self.engines)
        t = torch.tensor([n], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        if int(t.item()):
            print("rank %d: data-parallel step: %d flag wait(s) timed out here, %d on the worst rank -- no update has been "
                  "applied there since; stopping every rank" % (self.rank, n, int(t.item())), file=sys.stderr, flush=True)
            os._exit(5)

    def prepare(self, engine):
        """Attach the exchange to an engine; returns its step function."""
        if not self.active:
            return engine.step
        self.engines.append(engine)
        if self.comm is not None:
            # all-reduce of two gradient buckets and the full update on every rank; payload fp32 (the exact mean) by
            # default, RV_DDP_PAYLOAD=bf16 opts into the half-size payload (DESIGN.md section 5 has its error model)
            engine.attach_comm(self.comm, payload=os.environ.get("RV_DDP_PAYLOAD"))
            # RV_DDP_DEFER=1 (what bench.py times at N > 1): a step's last wait + update go out behind the next step's cast
            # launch; the engine's readers, its health check and the checkpoint path flush it themselves
            if os.environ.get("RV_DDP_DEFER", "0") == "1":
                engine.set_ddp_defer(True)
            return lambda x: engine.step_ddp(x, stream=torch.cuda.current_stream())
        from rawaudiovae_kelsey_amd import ddp
        sync = ddp.GradSync(engine.grad, ddp.engine_buckets(engine))
        return lambda x: ddp.ddp_step(engine, sync, x)

    def mean(self, value):
        """Mean over ranks of a host scalar."""
        if not self.active:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t)
        return float(t.item()) / self.world

    def check_replicas(self, engine):
        if not self.active:
            return
        chk = torch.stack([engine.param.double().sum(), engine.param.double().abs().sum()])
        lo, hi = chk.clone(), chk.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("data-parallel replicas diverged: %r vs %r" % (lo.tolist(), hi.tolist()))

    def close(self):
        if self.active:
            torch.cuda.synchronize()
            if self.comm is not None:
                self.comm.destroy()
            self.dist.barrier()
            self.dist.destroy_process_group()


def make_workspace(dataset, desc, run_number):
    """First free `<dataset>/<desc>/run-NNN`, counting up from run_number (train.py:94-107)."""
    run_id = run_number
    while True:
        workdir = dataset / desc / 'run-{:03d}'.format(run_id)
        try:
            os.makedirs(workdir)
            return workdir
        except FileExistsError:
            run_id += 1


def load_folder(folder, sampling_rate, verbose=True):
    """Concatenate every *.wav of `folder` in glob order (train.py:118-126)."""
    parts = []
    for f in folder.glob('*.wav'):
        if verbose:
            print('adding-> %s' % f.stem)
        parts.append(D.load_audio_mono(f, sampling_rate))
    if not parts:
        raise FileNotFoundError('no .wav files under {}'.format(folder.resolve()))
    return np.concatenate(parts, axis=0)


def init_test_audio(workdir, test_audio, test_folder, sampling_rate, segment_length, device):
    """rawvae/tests.py:13-42: list the test wavs, write test_original.wav, frame without overlap."""
    audio_log_dir = workdir / 'audio_logs'
    os.makedirs(audio_log_dir, exist_ok=True)
    test_files = [f for f in test_folder.glob('*.wav')]
    with open(audio_log_dir / (test_audio + '.txt'), 'w') as fh:
        fh.writelines("{}\n".format(f) for f in test_files)
    if not test_files:
        raise FileNotFoundError('no test .wav files under {}'.format(test_folder.resolve()))
    audio = np.concatenate([D.load_audio_mono(f, sampling_rate) for f in test_files], axis=0)
    D.write_wav(audio_log_dir / 'test_original.wav', audio, sampling_rate)
    return D.DeviceEvalAudio(audio, segment_length, device), audio_log_dir


def reconstruct(model, test_dataset, batch_size):
    """`model(test_sample)[0]` over the test set (train.py:218-232), on the GPU."""
    preds = []
    with torch.no_grad():
        for sample in test_dataset.batches(batch_size):
            preds.append(model(sample)[0])
    return torch.cat(preds, 0).view(-1).cpu().numpy()


class Writer:
    """SummaryWriter when tensorboard is importable, otherwise a no-op with the same calls."""

    def __init__(self, log_dir, enabled=True):
        self.w = None
        if enabled:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.w = SummaryWriter(log_dir=log_dir)
            except Exception:
                print('tensorboard not available: scalar/audio logging disabled')

    def __getattr__(self, name):
        if self.w is not None:
            return getattr(self.w, name)
        return lambda *a, **k: None


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='./default.ini', help='path to the config file')
    args = parser.parse_args(argv)
    config = read_config(args.config)

    sampling_rate = config['audio'].getint('sampling_rate')
    hop_length = config['audio'].getint('hop_length')
    segment_length = config['audio'].getint('segment_length')

    dataset = Path(config['dataset'].get('datapath'))
    if not dataset.exists():
        raise FileNotFoundError(dataset.resolve())
    run_number = config['dataset'].getint('run_number')
    my_audio = dataset / 'audio'
    test_audio = config['dataset'].get('test_dataset')
    dataset_test_audio = dataset / test_audio
    if not dataset_test_audio.exists():
        raise FileNotFoundError(dataset_test_audio.resolve())
    generate_test = config['dataset'].getboolean('generate_test')

    epochs = config['training'].getint('epochs')
    learning_rate = config['training'].getfloat('learning_rate')
    batch_size = config['training'].getint('batch_size')
    checkpoint_interval = config['training'].getint('checkpoint_interval')
    save_best_model_after = config['training'].getint('save_best_model_after')

    latent_dim = config['VAE'].getint('latent_dim')
    n_units = config['VAE'].getint('n_units')
    kl_beta = config['VAE'].getfloat('kl_beta')

    desc = config['extra'].get('description')
    start_time = time.time()
    config['extra']['start'] = time.asctime(time.localtime(start_time))

    hw = config['mi355x'] if config.has_section('mi355x') else {}
    seed = int(hw.get('seed', 0))
    ring = int(hw.get('loss_ring', 256))
    use_tb = str(hw.get('tensorboard', 'True')).lower() in ('1', 'true', 'yes')
    engine_kw = engine_options(hw)

    device = require_gpu(int(os.environ.get("LOCAL_RANK", "0")))
    dp = DataParallel(device)
    say = print if dp.main else (lambda *a, **k: None)
    device_name = torch.cuda.get_device_name()
    say('Device: {}'.format(device_name) + (' x {} (data parallel)'.format(dp.world) if dp.active else ''))
    config['VAE']['device_name'] = device_name

    workdir = dp.share(make_workspace(dataset, desc, run_number) if dp.main else None)
    config['dataset']['workspace'] = str(workdir.resolve())
    say("Workspace: {}".format(workdir))

    say('creating the dataset...')
    training_array = load_folder(my_audio, sampling_rate, verbose=dp.main)
    total_frames = len(training_array) // segment_length
    say('Total number of audio frames: {}'.format(total_frames))
    config['dataset']['total_frames'] = str(total_frames)
    training_dataset = D.DeviceAudio(training_array, segment_length, hop_length, device)
    n_batches = training_dataset.num_batches(batch_size * dp.world)

    config_path = workdir / 'config.ini'
    checkpoint_dir = workdir / 'model' / 'checkpoints'
    log_dir = workdir / 'logs'
    if dp.main:
        print("saving initial configs...")
        with open(config_path, 'w') as configfile:
            config.write(configfile)
        os.makedirs(checkpoint_dir, exist_ok=True)
        os.makedirs(log_dir, exist_ok=True)
    writer = Writer(log_dir, use_tb and dp.main)

    if generate_test and dp.main:
        test_dataset, audio_log_dir = init_test_audio(workdir, test_audio, dataset_test_audio, sampling_rate,
                                                      segment_length, device)

    torch.manual_seed(seed)   # same seed on every rank: identical initial replicas
    model = VAE(segment_length, n_units, latent_dim).to(device)
    n_frames = len(training_dataset)
    if n_frames < dp.world:
        raise ValueError("{} frames cannot be split over {} ranks".format(n_frames, dp.world))
    full = min(batch_size, n_frames // dp.world)                 # per-rank batch
    ragged = (n_frames % (full * dp.world)) // dp.world          # per-rank size of the epoch's last step
    engine = TrainEngine(segment_length, n_units, latent_dim, full, device=device, kl_beta=kl_beta,
                         lr=learning_rate, seed=seed + dp.rank, ring=ring, **engine_kw)   # per-rank eps stream
    engine.adopt(model)
    tail_engine = TrainEngine(segment_length, n_units, latent_dim, ragged, device=device, kl_beta=kl_beta,
                              lr=learning_rate, seed=seed + dp.rank, share=engine, **engine_kw) if ragged else None
    step_full = dp.prepare(engine)
    step_tail = dp.prepare(tail_engine) if tail_engine is not None else None
    # the data-parallel step forks its collectives from the caller's stream: it needs a non-default one
    train_stream = torch.cuda.Stream(device) if dp.active else None
    if train_stream is not None:
        # the engines' arenas, shadows and workspaces were filled on the default stream, which a fresh
        # stream does not wait for
        train_stream.wait_stream(torch.cuda.current_stream(device))
    # the same permutation on every rank: a device generator with the same seed (the permutation is drawn on the GPU;
    # torch's CPU randperm costs more than an epoch of training steps at 2e5 frames)
    shuffle_gen = torch.Generator(device=device).manual_seed(seed)

    def checkpoint_state(epoch):
        return {'epoch': epoch, 'state_dict': model.state_dict(), 'optimizer': engine.optimizer_state_dict()}

    def write_reconstruction(tag, epoch):
        audio_out = audio_log_dir / 'test_reconst_{:05d}.wav'.format(tag)
        pred = reconstruct(model, test_dataset, batch_size)
        D.write_wav(audio_out, pred, sampling_rate)
        print('Audio examples generated: {}'.format(audio_out))
        writer.add_audio('Reconstructed Audio', pred, epoch, sample_rate=sampling_rate)

    best_loss = float('inf')
    final_loss = float('inf')
    train_loss = 0.0
    epoch = 0
    for epoch in range(epochs):
        say('Epoch {}/{}'.format(epoch, epochs - 1))
        say('-' * 10)
        model.train()
        train_loss = 0.0
        seen = 0

        def drain():
            nonlocal train_loss, seen
            dp.check()      # (every attached engine, agreed across ranks, before anything is read back)
            for v in engine.drain_losses():
                writer.add_scalar('Loss/Batch', v, epoch * n_batches + seen)
                writer.add_scalar('Learning Rate', learning_rate, epoch * n_batches + seen)
                train_loss += v
                seen += 1

        with torch.cuda.stream(train_stream) if train_stream is not None else contextlib.nullcontext():
            if dp.active:
                epoch_batches = training_dataset.sharded_batches(full, dp.rank, dp.world, shuffle=True, generator=shuffle_gen)
                for i, batch in enumerate(epoch_batches):
                    (step_full if batch.shape[0] == full else step_tail)(batch)
                    if (i + 1) % ring == 0:
                        drain()
            else:
                # one GPU: the step reads its frames where the waveform lives (rv_plan_step_frames): the epoch is a
                # shuffled list of frame numbers, no framed copy of a batch is written
                for i, idx in enumerate(training_dataset.index_batches(batch_size, shuffle=True, generator=shuffle_gen)):
                    (engine if idx.numel() == full else tail_engine).step_frames(training_dataset, idx)
                    if (i + 1) % ring == 0:
                        drain()
            drain()
        if train_stream is not None:
            train_stream.synchronize()     # evaluation / checkpoints below read the parameters on the default stream
        train_loss = dp.mean(train_loss)   # global-batch loss = mean of the ranks' batch losses

        say('====> Epoch: {} - Total loss: {} - Average loss: {:.9f}'.format(
            epoch, train_loss, train_loss / len(training_dataset)))
        writer.add_scalar('Loss/train_total', train_loss, epoch)
        writer.add_scalar('Loss/train_average', train_loss / len(training_dataset), epoch)
        if dp.main:
            for name, param in model.named_parameters():
                writer.add_histogram(name, param, epoch)

        if epoch % checkpoint_interval == 0 and epoch != 0 and dp.main:
            print('Checkpoint - Epoch {}'.format(epoch))
            if generate_test:
                write_reconstruction(epoch, epoch)
            torch.save(checkpoint_state(epoch), checkpoint_dir / 'ckpt_{:05d}'.format(epoch))
            if train_loss < best_loss and epoch > save_best_model_after:
                save_path = workdir / 'model' / 'best_model.pt'
                torch.save(model, save_path)
                print('Epoch {:05d}: Saved {}'.format(epoch, save_path))
                config['training']['best_epoch'] = str(epoch)
                best_loss = train_loss
            elif train_loss > best_loss:
                print("Loss did not improve.")
        if epoch % checkpoint_interval == 0 and epoch != 0:
            dp.rendezvous()     # every rank: nobody starts the next epoch's steps while rank 0 is still writing
        final_loss = train_loss

    dp.check_replicas(engine)
    if dp.main:
        print('Last Checkpoint - Epoch {}'.format(epoch))
        if generate_test:
            write_reconstruction(epochs, epoch)
        torch.save(checkpoint_state(epoch), checkpoint_dir / 'ckpt_{:05d}'.format(epochs))

        if final_loss > best_loss:
            print("Final loss was not better than the last best model.")
            print("Final Loss: {}".format(final_loss))
            print("Best Loss: {}".format(best_loss))
        else:
            print("The last model is the best model.")
        torch.save(model, workdir / 'model' / 'last_model.pt')
        print('Training Finished: Saved the last model')

        config['extra']['end'] = time.asctime(time.localtime(time.time()))
        config['extra']['time_elapsed'] = str(time.time() - start_time)
        with open(config_path, 'w') as configfile:
            config.write(configfile)
    writer.close()
    dp.close()
    return workdir


if __name__ == '__main__':
    main()
