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

An optional `[mi355x]` section adds `seed`, `loss_ring` and `tensorboard` keys.
"""
import argparse
import configparser
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


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("train.py (MI355X build) needs a GPU: torch.cuda.is_available() is False")
    return torch.device('cuda')


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

    device = require_gpu()
    device_name = torch.cuda.get_device_name()
    print('Device: {}'.format(device_name))
    config['VAE']['device_name'] = device_name

    workdir = make_workspace(dataset, desc, run_number)
    config['dataset']['workspace'] = str(workdir.resolve())
    print("Workspace: {}".format(workdir))

    print('creating the dataset...')
    training_array = load_folder(my_audio, sampling_rate)
    total_frames = len(training_array) // segment_length
    print('Total number of audio frames: {}'.format(total_frames))
    config['dataset']['total_frames'] = str(total_frames)
    training_dataset = D.DeviceAudio(training_array, segment_length, hop_length, device)
    n_batches = training_dataset.num_batches(batch_size)

    print("saving initial configs...")
    config_path = workdir / 'config.ini'
    with open(config_path, 'w') as configfile:
        config.write(configfile)

    checkpoint_dir = workdir / 'model' / 'checkpoints'
    os.makedirs(checkpoint_dir, exist_ok=True)
    log_dir = workdir / 'logs'
    os.makedirs(log_dir, exist_ok=True)
    writer = Writer(log_dir, use_tb)

    if generate_test:
        test_dataset, audio_log_dir = init_test_audio(workdir, test_audio, dataset_test_audio, sampling_rate,
                                                      segment_length, device)

    torch.manual_seed(seed)
    model = VAE(segment_length, n_units, latent_dim).to(device)
    full = min(batch_size, len(training_dataset))
    engine = TrainEngine(segment_length, n_units, latent_dim, full, device=device, kl_beta=kl_beta,
                         lr=learning_rate, seed=seed, ring=ring)
    engine.adopt(model)
    ragged = len(training_dataset) % full
    tail_engine = TrainEngine(segment_length, n_units, latent_dim, ragged, device=device, kl_beta=kl_beta,
                              lr=learning_rate, seed=seed, share=engine) if ragged else None
    shuffle_gen = torch.Generator().manual_seed(seed)

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
        print('Epoch {}/{}'.format(epoch, epochs - 1))
        print('-' * 10)
        model.train()
        train_loss = 0.0
        seen = 0

        def drain():
            nonlocal train_loss, seen
            for v in engine.drain_losses():
                writer.add_scalar('Loss/Batch', v, epoch * n_batches + seen)
                writer.add_scalar('Learning Rate', learning_rate, epoch * n_batches + seen)
                train_loss += v
                seen += 1

        for i, batch in enumerate(training_dataset.batches(batch_size, shuffle=True, generator=shuffle_gen)):
            (engine if batch.shape[0] == full else tail_engine).step(batch)
            if (i + 1) % ring == 0:
                drain()
        drain()

        print('====> Epoch: {} - Total loss: {} - Average loss: {:.9f}'.format(
            epoch, train_loss, train_loss / len(training_dataset)))
        writer.add_scalar('Loss/train_total', train_loss, epoch)
        writer.add_scalar('Loss/train_average', train_loss / len(training_dataset), epoch)
        for name, param in model.named_parameters():
            writer.add_histogram(name, param, epoch)

        if epoch % checkpoint_interval == 0 and epoch != 0:
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
        final_loss = train_loss

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
    return workdir


if __name__ == '__main__':
    main()
