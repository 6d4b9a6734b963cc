"""Audio front-end on the GPU: waveform -> the (92, frames) log-frequency spectrogram the
spectrogram tower takes, i.e. the madmom processor chain of the reference (tutorials/Embedding Tutorial.ipynb cell 28;
`processor.process(audio_file).T` at audio_sheet_server.py:632,678, audio2sheet_align.py:99):

    SignalProcessor(num_channels=1, sample_rate=22050) -> FramedSignalProcessor(frame_size=2048, fps=20, origin='future')
    -> FilteredSpectrogramProcessor(LogarithmicFilterbank, num_bands=16, fmin=30, fmax=6000)
    -> LogarithmicSpectrogramProcessor()

Decoding / resampling audio files (ffmpeg inside madmom) stays with the caller: `process(samples)` takes mono
samples at SAMPLE_RATE.  The filterbank is built on the host with madmom's published construction (third-party
semantic, unverified offline; it reproduces the reference's 92 bands); framing, windowed DFT magnitudes,
filterbank and logarithm run in one kernel per call (csrc/piece_vote_kernels.hip: spectrogram_kernel).
"""
from __future__ import print_function

import numpy as np

SAMPLE_RATE = 22050
FRAME_SIZE = 2048
FPS = 20


def _log_frequencies(bands_per_octave, fmin, fmax, fref=440.0):
    left = np.floor(np.log2(float(fmin) / fref) * bands_per_octave)
    right = np.ceil(np.log2(float(fmax) / fref) * bands_per_octave)
    freqs = fref * 2.0 ** (np.arange(left, right) / float(bands_per_octave))
    freqs = freqs[np.searchsorted(freqs, fmin):]
    return freqs[:np.searchsorted(freqs, fmax, 'right')]


def _frequencies2bins(frequencies, bin_frequencies):
    idx = bin_frequencies.searchsorted(frequencies)
    idx = np.clip(idx, 1, len(bin_frequencies) - 1)
    left, right = bin_frequencies[idx - 1], bin_frequencies[idx]
    idx -= frequencies - left < right - frequencies
    return np.unique(idx)                                  # unique_filters=True


def logarithmic_filterbank(sample_rate=SAMPLE_RATE, frame_size=FRAME_SIZE, num_bands=16, fmin=30.0, fmax=6000.0):
    """madmom.audio.filters.LogarithmicFilterbank(norm_filters=True, unique_filters=True) as
    (starts int32[nf], lengths int32[nf], weights float32[sum lengths])."""
    bin_freqs = np.fft.fftfreq(frame_size, 1.0 / sample_rate)[:frame_size >> 1]
    bins = _frequencies2bins(_log_frequencies(num_bands, fmin, fmax), bin_freqs)
    starts, lens, weights = [], [], []
    for start, center, stop in zip(bins[:-2], bins[1:-1], bins[2:]):
        if stop - start < 2:
            center, stop = start, start + 1
        c = int(center - start)
        data = np.zeros(int(stop - start), dtype=np.float32)
        data[:c] = np.linspace(0, 1, c, endpoint=False)
        data[c:] = np.linspace(1, 0, int(stop - center), endpoint=False)
        data /= data.sum()
        starts.append(int(start))
        lens.append(len(data))
        weights.append(data)
    return np.asarray(starts, np.int32), np.asarray(lens, np.int32), np.concatenate(weights).astype(np.float32)


class SpectrogramProcessor(object):
    """processor = SequentialProcessor([sig_proc, fsig_proc, spec_proc, log_spec_proc]) of the reference, on the GPU."""

    def __init__(self, engine, sample_rate=SAMPLE_RATE, frame_size=FRAME_SIZE, fps=FPS, window_scale=1.0):
        self.engine = engine
        self.sample_rate, self.frame_size, self.fps = sample_rate, frame_size, fps
        self.hop = sample_rate / float(fps)
        # int16 input: madmom divides the window by the integer range (stft.py); pass window_scale = 1 / 32767
        self.window = (np.hanning(frame_size) * window_scale).astype(np.float32)
        self.fb_start, self.fb_len, self.fb_w = logarithmic_filterbank(sample_rate, frame_size)
        self.num_bins = len(self.fb_start)

    def num_frames(self, n_samples):
        return int(np.ceil(n_samples / float(self.hop)))

    def process_dev(self, samples):
        """-> (DeviceBuffer holding the (num_bins, n_frames) float32 spectrogram, n_frames)"""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        n = self.num_frames(samples.size)
        eng = self.engine
        d_in = eng.alloc(max(4, samples.nbytes)).upload(samples)
        d_out = eng.alloc(max(4, n * self.num_bins * 4))
        try:
            eng.spectrogram_dev(d_in.ptr, samples.size, self.frame_size, self.hop, self.window, self.fb_start,
                                self.fb_len, self.fb_w, n, d_out.ptr, transposed=True)
        finally:
            d_in.free()
        return d_out, n

    def process(self, samples):
        """the reference's `processor.process(audio).T`: (num_bins, n_frames) float32"""
        d_out, n = self.process_dev(samples)
        out = d_out.download((self.num_bins, n), np.float32)
        d_out.free()
        return out
