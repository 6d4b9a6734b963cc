"""TEST INFRASTRUCTURE (see oracle/__init__.py: parity unpinned by the reference).

CPU restatement of the audio front-end the reference uses (tutorials/Embedding Tutorial.ipynb cell 28;
msmd.midi_parser.processor, used at audio_sheet_server.py:632,678 and audio2sheet_align.py:99):
    SignalProcessor(num_channels=1, sample_rate=22050)
    FramedSignalProcessor(frame_size=2048, fps=20, origin='future')
    FilteredSpectrogramProcessor(LogarithmicFilterbank, num_bands=16, fmin=30, fmax=6000)
    LogarithmicSpectrogramProcessor()          # log10(1 + x)
madmom (third party, pinned madmom==0.15.1 in requirements.txt) is absent here: its algorithm is restated from the
published source - audio/signal.py (signal_frame, FramedSignal), audio/stft.py (stft, fft_frequencies),
audio/filters.py (log_frequencies, frequencies2bins, TriangularFilter, LogarithmicFilterbank).  Unverified offline,
with ONE anchor inside the reference: this construction yields exactly the 92 bands the reference hard-codes
(utils/data_pools.py:18 SPEC_BINS = 92, models INPUT_SHAPE_2 = [1, 92, 42]); tests/test_audio_frontend.py checks it.
"""
import numpy as np

SAMPLE_RATE, FRAME_SIZE, FPS = 22050, 2048, 20


def log_frequencies(bands_per_octave, fmin, fmax, fref=440.0):
    left = np.floor(np.log2(float(fmin) / fref) * bands_per_octave)
    right = np.ceil(np.log2(float(fmax) / fref) * bands_per_octave)
    frequencies = fref * 2.0 ** (np.arange(left, right) / float(bands_per_octave))
    frequencies = frequencies[np.searchsorted(frequencies, fmin):]
    frequencies = frequencies[:np.searchsorted(frequencies, fmax, 'right')]
    return frequencies


def frequencies2bins(frequencies, bin_frequencies, unique_bins=False):
    frequencies = np.asarray(frequencies)
    bin_frequencies = np.asarray(bin_frequencies)
    indices = bin_frequencies.searchsorted(frequencies)
    indices = np.clip(indices, 1, len(bin_frequencies) - 1)
    left = bin_frequencies[indices - 1]
    right = bin_frequencies[indices]
    indices -= frequencies - left < right - frequencies
    if unique_bins:
        indices = np.unique(indices)
    return indices


def logarithmic_filterbank(sample_rate=SAMPLE_RATE, frame_size=FRAME_SIZE, num_bands=16, fmin=30.0, fmax=6000.0,
                           norm_filters=True):
    """-> (starts, filters): triangular filter f covers FFT bins starts[f] .. starts[f] + len(filters[f])"""
    num_fft_bins = frame_size >> 1
    bin_frequencies = np.fft.fftfreq(frame_size, 1.0 / sample_rate)[:num_fft_bins]
    frequencies = log_frequencies(num_bands, fmin, fmax)
    bins = frequencies2bins(frequencies, bin_frequencies, unique_bins=True)
    starts, filters = [], []
    for start, center, stop in zip(bins[:-2], bins[1:-1], bins[2:]):
        if stop - start < 2:                       # too small: one-bin filter (TriangularFilter.band_bins)
            center = start
            stop = start + 1
        center_rel = int(center - start)
        data = np.zeros(int(stop - start), dtype=np.float32)
        data[:center_rel] = np.linspace(0, 1, center_rel, endpoint=False)
        data[center_rel:] = np.linspace(1, 0, int(stop - center), endpoint=False)
        if norm_filters:
            data /= data.sum()
        starts.append(int(start))
        filters.append(data)
    return starts, filters


def num_frames(n_samples, hop):
    return int(np.ceil(n_samples / float(hop)))


def spectrogram(samples, sample_rate=SAMPLE_RATE, frame_size=FRAME_SIZE, fps=FPS, window_scale=1.0):
    """processor.process(...) -> (n_frames, 92) float32; the reference transposes it (`.T`)."""
    samples = np.asarray(samples, dtype=np.float32)
    hop = sample_rate / float(fps)
    n = num_frames(len(samples), hop)
    window = (np.hanning(frame_size) * window_scale).astype(np.float32)
    starts, filters = logarithmic_filterbank(sample_rate, frame_size)
    out = np.zeros((n, len(filters)), dtype=np.float32)
    for i in range(n):
        start = int(i * hop)                       # origin 'future': the frame begins at the reference sample
        frame = np.zeros(frame_size, dtype=np.float32)
        seg = samples[start:start + frame_size]
        frame[:len(seg)] = seg
        spec = np.abs(np.fft.fft(frame * window)[:frame_size >> 1]).astype(np.float32)
        for f, (s, w) in enumerate(zip(starts, filters)):
            out[i, f] = np.dot(spec[s:s + len(w)], w)
    return np.log10(out + 1.0).astype(np.float32)
