#!/usr/bin/env python3
"""`python train_iterable.py --config kelsey_iterable.ini` -- the reference's streaming entry
point (/root/reference/train_iterable.py) on the MI355X path.

Epoch-less training: `total_num_batches = int(total_num_frames / batch_size)` batches are drawn
from an endless stream of hop-strided frames (file list shuffled once, cycled; channel 0;
train_iterable.py:70-74,195; rawvae/dataset.py:38-84), checkpoints are indexed by batch
(`ckpt_%05d` with key `batch_id`, train_iterable.py:220-226), stdout is teed to
`<workdir>/console_log` (117-133) and every batch prints `====> Batch: i - Loss: x`.

Differences from the reference: the step is `TrainEngine.step`; frames are gathered on the
device from per-file waveforms cached in HBM; batch losses are printed when the device loss
ring is drained (every `loss_ring` batches) rather than with a sync per batch; the frame length
follows `[audio] segment_length` (the reference hard-codes 1024, dataset.py:66).
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import train as T  # noqa: E402  (shared helpers: config, workspace, test audio, writer)
from rawvae.model import VAE  # noqa: E402
from rawaudiovae_kelsey_amd import data as D  # noqa: E402
from rawaudiovae_kelsey_amd.engine import TrainEngine  # noqa: E402


class Tee:
    """Duplicate stdout into a file (train_iterable.py:117-133)."""

    def __init__(self, path):
        self.file = open(path, 'w')
        self.stdout = sys.stdout

    def write(self, data):
        self.stdout.write(data)
        self.file.write(data)
        self.file.flush()

    def flush(self):
        self.stdout.flush()
        self.file.flush()

    def close(self):
        sys.stdout = self.stdout
        self.file.close()


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', type=str, default='./default_iterable.ini', help='path to the config file')
    args = parser.parse_args(argv)
    config = T.read_config(args.config)

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

    total_num_frames = config['training'].getint('total_num_frames')
    learning_rate = config['training'].getfloat('learning_rate')
    batch_size = config['training'].getint('batch_size')
    checkpoint_interval = config['training'].getint('checkpoint_interval')
    total_num_batches = int(total_num_frames / batch_size)
    latent_dim = config['VAE'].getint('latent_dim')
    n_units = config['VAE'].getint('n_units')
    kl_beta = config['VAE'].getfloat('kl_beta')
    desc = config['extra'].get('description')
    start_time = time.time()
    config['extra']['start'] = time.asctime(time.localtime(start_time))
    hw = config['mi355x'] if config.has_section('mi355x') else {}
    seed = int(hw.get('seed', 0))
    ring = int(hw.get('loss_ring', 64))
    use_tb = str(hw.get('tensorboard', 'True')).lower() in ('1', 'true', 'yes')

    device = T.require_gpu()
    device_name = torch.cuda.get_device_name()
    print('Device: {}'.format(device_name))
    config['VAE']['device_name'] = device_name

    workdir = T.make_workspace(dataset, desc, run_number)
    config['dataset']['workspace'] = str(workdir.resolve())
    tee = Tee(workdir / 'console_log')
    sys.stdout = tee
    try:
        print("Workspace: {}".format(workdir))
        print('creating the dataset...')
        files = sorted(my_audio.glob('*.wav'))
        stream = D.StreamingFrames(files, sampling_rate, hop_length, segment_length, device, shuffle=True, seed=seed)
        print('Total number of batches: {}'.format(total_num_batches))

        config_path = workdir / 'config.ini'
        with open(config_path, 'w') as configfile:
            config.write(configfile)
        checkpoint_dir = workdir / 'model' / 'checkpoints'
        os.makedirs(checkpoint_dir, exist_ok=True)
        log_dir = workdir / 'logs'
        os.makedirs(log_dir, exist_ok=True)
        writer = T.Writer(log_dir, use_tb)
        if generate_test:
            test_dataset, audio_log_dir = T.init_test_audio(workdir, test_audio, dataset_test_audio, sampling_rate,
                                                            segment_length, device)

        torch.manual_seed(seed)
        model = VAE(segment_length, n_units, latent_dim).to(device)
        engine = TrainEngine(segment_length, n_units, latent_dim, batch_size, device=device, kl_beta=kl_beta,
                             lr=learning_rate, seed=seed, ring=ring, **T.engine_options(hw))
        engine.adopt(model)
        model.train()

        train_loss = 0.0
        best_loss = float('inf')
        logged = 0

        def drain():
            nonlocal train_loss, logged
            for v in engine.drain_losses():
                writer.add_scalar('Loss/Batch', v, logged)
                writer.add_scalar('Learning Rate', learning_rate, logged)
                print('====> Batch: {} - Loss: {:.9f}'.format(logged, v))
                train_loss += v
                logged += 1

        batch_id = 0
        for batch_id, data in enumerate(stream.batches(batch_size, total_num_batches)):
            engine.step(data)
            if (batch_id + 1) % ring == 0:
                drain()
            if batch_id % checkpoint_interval == 0 and batch_id != 0:
                drain()
                print('Checkpoint - Epoch {}'.format(batch_id))
                state = {'batch_id': batch_id, 'state_dict': model.state_dict(),
                         'optimizer': engine.optimizer_state_dict()}
                if generate_test:
                    audio_out = audio_log_dir / 'test_reconst_{:05d}.wav'.format(batch_id)
                    pred = T.reconstruct(model, test_dataset, batch_size)
                    D.write_wav(audio_out, pred, sampling_rate)
                    print('Audio examples generated: {}'.format(audio_out))
                    writer.add_audio('Reconstructed Audio', pred, batch_id, sample_rate=sampling_rate)
                torch.save(state, checkpoint_dir / 'ckpt_{:05d}'.format(batch_id))
                if train_loss < best_loss:
                    save_path = workdir / 'model' / 'best_model.pt'
                    torch.save(model, save_path)
                    print('batch_id {:05d}: Saved {}'.format(batch_id, save_path))
                    config['training']['best_model'] = str(batch_id)
                    best_loss = train_loss
                else:
                    print("Loss did not improve.")
        drain()

        # the reference increments batch_id at the end of every iteration (train_iterable.py:266-267), so
        # its final checkpoint carries batch_id == total_num_batches
        batch_id = total_num_batches
        print('Last Checkpoint - batch_id {}'.format(batch_id))
        state = {'batch_id': batch_id, 'state_dict': model.state_dict(), 'optimizer': engine.optimizer_state_dict()}
        if generate_test:
            audio_out = audio_log_dir / 'test_reconst_{:05d}.wav'.format(total_num_batches)
            pred = T.reconstruct(model, test_dataset, batch_size)
            D.write_wav(audio_out, pred, sampling_rate)
            print('Last Audio examples generated: {}'.format(audio_out))
        torch.save(state, checkpoint_dir / 'ckpt_{:05d}'.format(total_num_batches))
        torch.save(model, workdir / 'model' / 'last_model.pt')
        print('Training Finished: Saved the last model')
        config['extra']['end'] = time.asctime(time.localtime(time.time()))
        config['extra']['time_elapsed'] = str(time.time() - start_time)
        with open(config_path, 'w') as configfile:
            config.write(configfile)
        writer.close()
    finally:
        tee.close()
    return workdir


if __name__ == '__main__':
    main()
