"""Audio framing for the training path, device-resident.

Semantics follow the reference's datasets (`rawvae/dataset.py`):
  * `AudioDataset` (dataset.py:86-121): concatenate all training wavs, zero-pad to a multiple
    of `hop`, frame i = audio[i*hop : i*hop + S], len = N/hop - S/hop + 1, `ValueError` when
    S is not a multiple of hop; `DataLoader(shuffle=True)` draws a fresh permutation per epoch
    and keeps the ragged last batch (train.py:134).
  * `TestDataset` (dataset.py:129-160): non-overlapping frames, tail zero-padded to S.
  * `IterableAudioDataset` (dataset.py:11-84): shuffle the FILE list once per iterator, cycle
    it forever, per file take channel 0, pad to `hop`, emit hop-strided frames in order.

A Python DataLoader delivers ~0.1 M frames/s; the step consumes >15 M frames/s, so
the waveform is uploaded once and frames are gathered on the GPU (`rv_gather_frames`): frames
overlap S/hop-fold, so the unique bytes are 1/8 of the framed batch at hop 128.

wav I/O uses scipy (librosa / soundfile / torchaudio are not available here): PCM is scaled to
[-1, 1] float32; `librosa.load(sr=...)`'s mono mix-down (mean of channels) and resampling
(polyphase here, not librosa's soxr) are restated; the streaming path's `torchaudio.functional.resample` is restated from
its published algorithm (`_resample_sinc_hann`).
"""
import itertools
import random

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, stream_ptr


def _to_float32(a):
    if a.dtype == np.float32:
        return a
    if a.dtype == np.float64:
        return a.astype(np.float32)
    if a.dtype == np.uint8:
        return (a.astype(np.float32) - 128.0) / 128.0
    if np.issubdtype(a.dtype, np.integer):
        return a.astype(np.float32) / float(2 ** (8 * a.dtype.itemsize - 1))
    raise ValueError("unsupported wav sample type %s" % a.dtype)


def read_wav(path):
    """-> (float32 [n] or [n, channels], sample_rate)."""
    from scipy.io import wavfile
    sr, a = wavfile.read(str(path))
    return _to_float32(a), int(sr)


def _resample(a, sr_in, sr_out):
    if sr_in == sr_out:
        return a
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(a, sr_out // g, sr_in // g).astype(np.float32)


def _resample_sinc_hann(a, sr_in, sr_out, lowpass_filter_width=6, rolloff=0.99, chunk=1 << 18):
    """`torchaudio.functional.resample(waveform, orig_freq, new_freq)` with its defaults (resampling_method
    "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99), the call of dataset.py:50-51, restated from the
    published algorithm of torchaudio 2.x (`torchaudio/functional/functional.py`: `_get_sinc_resample_kernel` +
    `_apply_sinc_resample_kernel`) -- torchaudio is not installed here, so no fixture pins this ("parity unpinned",
    DESIGN.md section 4; tests check it against the interpolation formula it implements).  With orig / new the two rates
    over their gcd: a bank of `new` Hann-windowed sinc filters of 2 * width + orig taps, cut-off rolloff * min(orig, new)
    / 2, applied with stride orig to the signal padded by (width, width + orig) zeros; output j * new + i is filter i at
    input frame j; ceil(new * n / orig) samples are kept.  The filter bank is evaluated in float64 and rounded to
    float32, the convolution runs in float32, as torchaudio's does."""
    import math
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    if orig == new:
        return a
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    # (torch divides the int64 phase indices by new_freq in float32 before they meet the float64 tap positions)
    t = (np.arange(0, -new, -1).astype(np.float32) / np.float32(new)).astype(np.float64)[:, None] + idx
    t *= base
    np.clip(t, -lowpass_filter_width, lowpass_filter_width, out=t)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    with np.errstate(divide="ignore", invalid="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels = (kernels * window * (base / orig)).astype(np.float32)          # [new, 2 * width + orig]
    taps = kernels.shape[1]
    n = len(a)
    x = np.zeros(n + 2 * width + orig, np.float32)
    x[width:width + n] = a
    n_frames = (len(x) - taps) // orig + 1
    out = np.empty((n_frames, new), np.float32)
    kt = np.ascontiguousarray(kernels.T)
    for f0 in range(0, n_frames, chunk):
        f1 = min(n_frames, f0 + chunk)
        seg = x[f0 * orig:(f1 - 1) * orig + taps]
        frames = np.lib.stride_tricks.as_strided(seg, shape=(f1 - f0, taps), strides=(orig * seg.itemsize, seg.itemsize),
                                                 writeable=False)
        np.matmul(frames, kt, out=out[f0:f1])
    target = int(math.ceil(new * n / orig))
    return out.reshape(-1)[:target]


def load_audio_mono(path, sampling_rate):
    """`librosa.load(path, sr=sampling_rate)` (train.py:120): mono float32 at `sampling_rate`."""
    a, sr = read_wav(path)
    if a.ndim == 2:
        a = a.mean(axis=1).astype(np.float32)
    return _resample(a, sr, sampling_rate)


def load_audio_ch0(path, sampling_rate):
    """`torchaudio.load` + first channel only (dataset.py:47-58)."""
    a, sr = read_wav(path)
    if a.ndim == 2:
        a = np.ascontiguousarray(a[:, 0])
    return _resample_sinc_hann(a, sr, sampling_rate)


def write_wav(path, samples, sampling_rate):
    """`soundfile.write(path, samples, sr)` stand-in: float32 wav."""
    from scipy.io import wavfile
    wavfile.write(str(path), int(sampling_rate), np.asarray(samples, dtype=np.float32))


def frame_count(n_samples, segment_length, hop):
    """(number of frames, padded length) of AudioDataset (dataset.py:99-104,121)."""
    if segment_length % hop != 0:
        raise ValueError("segment_length {} is not a multiple of hop_size {}".format(segment_length, hop))
    padded = n_samples if n_samples % hop == 0 else n_samples + hop - n_samples % hop
    return padded // hop - segment_length // hop + 1, padded


class DeviceAudio:
    """AudioDataset with the waveform resident in HBM and frames gathered on the device."""

    def __init__(self, audio_np, segment_length, hop_size, device="cuda"):
        self.segment_length, self.hop_size = int(segment_length), int(hop_size)
        self.n_frames, self.padded = frame_count(len(audio_np), self.segment_length, self.hop_size)
        self.device = torch.device(device)
        buf = np.zeros(self.padded, dtype=np.float32)
        buf[:len(audio_np)] = audio_np
        self.audio = torch.from_numpy(buf).to(self.device)
        # The same waveform as bf16, cast ONCE here (round to nearest even: what the per-batch cast kernel did to every
        # frame), followed by zeros: fc1's GEMM stages its operand tiles straight from it (rv_linear_fwd_frames), and a
        # frame's padded tail (segment length rounded up to the tile grid) reads past the last sample.
        slack = (self.segment_length + 127) // 128 * 128 + 8
        self.audio_bf16 = torch.zeros(self.padded + slack, dtype=torch.bfloat16, device=self.device)
        self.audio_bf16[:self.padded].copy_(self.audio)

    def __len__(self):
        return max(self.n_frames, 0)

    def num_batches(self, batch_size):
        return (len(self) + batch_size - 1) // batch_size

    def _perm(self, n, shuffle, generator):
        """The epoch's frame order (DataLoader(shuffle=True) draws a fresh permutation per epoch, train.py:134).
        Drawn ON THE DEVICE unless a CPU generator is passed: torch's CPU randperm takes ~50 ms for 2e5 frames,
        several epochs' worth of training steps."""
        if not shuffle:
            return torch.arange(n, device=self.device)
        if generator is not None and generator.device.type == "cpu":
            return torch.randperm(n, generator=generator).to(self.device)
        return torch.randperm(n, device=self.device, generator=generator)

    def gather(self, index, out=None, stream=None):
        """index: int64 device tensor of frame numbers -> fp32 [len(index), S]."""
        n = index.numel()
        if out is None:
            out = torch.empty((n, self.segment_length), dtype=torch.float32, device=self.device)
        lib().rv_gather_frames(ptr(self.audio), self.padded, ptr(index), 0, n, self.segment_length,
                               self.hop_size, ptr(out), stream_ptr(stream))
        return out

    def frames(self, first, n, out=None, stream=None):
        """n consecutive frames starting at frame `first`."""
        if out is None:
            out = torch.empty((n, self.segment_length), dtype=torch.float32, device=self.device)
        lib().rv_gather_frames(ptr(self.audio), self.padded, None, first, n, self.segment_length,
                               self.hop_size, ptr(out), stream_ptr(stream))
        return out

    def batches(self, batch_size, shuffle=True, generator=None):
        """One epoch: DataLoader(dataset, batch_size, shuffle) -- fresh permutation, ragged last batch kept."""
        n = len(self)
        perm = self._perm(n, shuffle, generator)
        for lo in range(0, n, batch_size):
            yield self.gather(perm[lo:lo + batch_size])


    def index_batches(self, batch_size, shuffle=True, generator=None):
        """One epoch as device int64 index tensors (what `batches` gathers): for `TrainEngine.step_frames`, which
        reads the frames where the waveform lives instead of from a gathered copy."""
        n = len(self)
        perm = self._perm(n, shuffle, generator)
        for lo in range(0, n, batch_size):
            yield perm[lo:lo + batch_size].contiguous()

    def sharded_batches(self, batch_size, rank, world, shuffle=True, generator=None):
        """One data-parallel epoch.  The epoch's permutation (the same `generator` state on every rank) is
        cut into global batches of world * batch_size frames and rank r takes the r-th slice of each; the
        ragged tail is split evenly, because every rank must step with the same batch size for the mean of
        rank gradients to be the global-batch gradient (up to world - 1 frames of an epoch are left out)."""
        n = len(self)
        perm = self._perm(n, shuffle, generator)
        gb = world * batch_size
        full = n // gb
        for i in range(full):
            lo = i * gb + rank * batch_size
            yield self.gather(perm[lo:lo + batch_size])
        tail = (n - full * gb) // world
        if tail:
            lo = full * gb + rank * tail
            yield self.gather(perm[lo:lo + tail])


class DeviceEvalAudio(DeviceAudio):
    """TestDataset: non-overlapping frames, tail zero-padded to a whole frame."""

    def __init__(self, audio_np, segment_length, device="cuda"):
        n = len(audio_np)
        pad = n if n % segment_length == 0 else n + segment_length - n % segment_length
        a = np.zeros(pad, dtype=np.float32)
        a[:n] = audio_np
        super().__init__(a, segment_length, segment_length, device)

    def batches(self, batch_size, shuffle=False, generator=None):
        return super().batches(batch_size, shuffle=False)


class StreamingFrames:
    """IterableAudioDataset + DataLoader(batch_size, shuffle=False) + islice: an endless stream of
    fixed-size batches of hop-strided 1024-sample... `segment_length`-sample frames, file order
    shuffled once per iterator (dataset.py:38-42,77-84).  Each file's waveform is uploaded once
    and cached on the device; a batch that straddles a file boundary is gathered in two pieces."""

    def __init__(self, files, sampling_rate, hop_size, segment_length, device="cuda", shuffle=True, seed=None,
                 cache_bytes=8 << 30):
        self.files = list(files)
        if not self.files:
            raise FileNotFoundError("no .wav files to stream")
        self.sampling_rate, self.hop_size, self.segment_length = int(sampling_rate), int(hop_size), int(segment_length)
        self.device = torch.device(device)
        self.shuffle = shuffle
        self._rng = random.Random(seed)
        # decoded waveforms kept on the device, least recently used first out once `cache_bytes` is exceeded
        # (the reference streams files lazily, dataset.py:44-75; a corpus larger than HBM must not pile up here)
        self._cache, self._cache_bytes, self.cache_limit = {}, 0, int(cache_bytes)

    def _dataset(self, f):
        d = self._cache.pop(f, None)
        if d is None:
            d = DeviceAudio(load_audio_ch0(f, self.sampling_rate), self.segment_length, self.hop_size, self.device)
            self._cache_bytes += d.audio.numel() * 4
            while self._cache and self._cache_bytes > self.cache_limit:
                old = self._cache.pop(next(iter(self._cache)))
                self._cache_bytes -= old.audio.numel() * 4
        self._cache[f] = d   # most recently used last
        return d

    def batches(self, batch_size, n_batches):
        order = self._rng.sample(self.files, len(self.files)) if self.shuffle else list(self.files)
        stream = itertools.cycle(order)
        cur, pos = None, 0
        empty_run = 0   # consecutive files without a single full frame
        for _ in range(n_batches):
            out = torch.empty((batch_size, self.segment_length), dtype=torch.float32, device=self.device)
            filled = 0
            while filled < batch_size:
                if cur is None or pos >= len(cur):
                    cur, pos = self._dataset(next(stream)), 0
                    if len(cur) <= 0:
                        cur = None
                        empty_run += 1
                        if empty_run >= len(order):
                            raise ValueError("no file yields a full %d-sample frame" % self.segment_length)
                        continue
                    empty_run = 0
                take = min(batch_size - filled, len(cur) - pos)
                cur.frames(pos, take, out=out[filled:filled + take])
                filled += take
                pos += take
            yield out
