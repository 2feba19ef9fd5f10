"""Host-side signal-quality metrics (numpy / scipy), restating the reference's utils/metrics.py:
NMSE (:42-52), EVM (:55-108), ACLR (:111-151), power_spectrum via scipy.signal.welch (:154-187).
They run once per epoch on (segments, nperseg, 2) arrays, so they stay on the host."""
import numpy as np


def IQ_to_complex(iq):
    return iq[..., 0] + 1j * iq[..., 1]


def NMSE(prediction, ground_truth):
    """Mean over segments of 10 log10( mean|e|^2 / mean|truth|^2 )."""
    err = np.square(ground_truth[..., 0] - prediction[..., 0]) + np.square(ground_truth[..., 1] - prediction[..., 1])
    energy = np.square(ground_truth[..., 0]) + np.square(ground_truth[..., 1])
    return np.mean(10 * np.log10(np.mean(err, axis=-1) / np.mean(energy, axis=-1)))


def _main_channel_bins(freq, bw_main_ch, n_sub_ch):
    lo = int(np.min(np.where(freq >= -bw_main_ch / 2)))
    hi = int(np.max(np.where(freq <= bw_main_ch / 2)))
    return lo, hi, int((hi - lo) / n_sub_ch)


def EVM(prediction, ground_truth, sample_rate=int(800e6), bw_main_ch=200e6, n_sub_ch=10, nperseg=2560):
    """20 log10 of the mean (over segments and sub-channels) relative spectral error inside the main channel."""
    sp = np.fft.fftshift(np.fft.fft(IQ_to_complex(prediction), n=nperseg, axis=-1), axes=-1)
    sg = np.fft.fftshift(np.fft.fft(IQ_to_complex(ground_truth), n=nperseg, axis=-1), axes=-1)
    freq = np.fft.fftshift(np.fft.fftfreq(prediction.shape[1], d=1 / sample_rate))
    lo, _, w = _main_channel_bins(freq, bw_main_ch, n_sub_ch)
    err = np.zeros((prediction.shape[0], n_sub_ch))
    for c in range(n_sub_ch):      # as metrics.py:92-103: the complex64 means land in the float64 array BEFORE the division
        sl = slice(lo + c * w, lo + (c + 1) * w)
        err[:, c] = np.mean(np.abs(sp[:, sl] - sg[:, sl]), axis=-1)
        err[:, c] = err[:, c] / np.mean(np.abs(sg[:, sl]), axis=-1)
    return 20 * np.log10(np.mean(err.mean(axis=-1)))


def power_spectrum(complex_signal, fs=800e6, nperseg=2560, axis=-1):
    """Two-sided Welch spectrum ('spectrum' scaling), fft-shifted, averaged over segments."""
    from scipy.signal import welch
    freq, ps = welch(complex_signal, fs=fs, nperseg=nperseg, return_onesided=False, scaling="spectrum", axis=-1)
    half = int(nperseg / 2)
    freq = np.concatenate((freq[half:], freq[:half]))
    ps = np.concatenate((ps[..., half:], ps[..., :half]), axis=-1)
    return freq, np.mean(ps, axis=0)


def ACLR(prediction, fs=800e6, nperseg=2560, bw_main_ch=200e6, n_sub_ch=10):
    """(left, right) adjacent-channel power relative to the strongest main sub-channel, in dB."""
    freq, psd = power_spectrum(IQ_to_complex(prediction), fs=fs, nperseg=nperseg, axis=-1)
    lo, hi, w = _main_channel_bins(freq, bw_main_ch, n_sub_ch)
    sub = np.zeros(n_sub_ch)       # float64 like metrics.py:137-141: the ratio and its log are then taken in double
    for c in range(n_sub_ch):
        sub[c] = np.sum(psd[lo + c * w:lo + (c + 1) * w])
    ref = sub.max()
    left = np.mean(10 * np.log10(np.sum(psd[lo - w:lo]) / ref))
    right = np.mean(10 * np.log10(np.sum(psd[hi:hi + w]) / ref))
    return left, right


def calculate_metrics(args, stat, prediction, ground_truth):
    """modules/train_funcs.py:93-105."""
    stat["NMSE"] = NMSE(prediction, ground_truth)
    stat["EVM"] = EVM(prediction, ground_truth, bw_main_ch=args.bw_main_ch, n_sub_ch=args.n_sub_ch, nperseg=args.nperseg)
    left, right = ACLR(prediction, fs=args.input_signal_fs, nperseg=args.nperseg, bw_main_ch=args.bw_main_ch,
                       n_sub_ch=args.n_sub_ch)
    stat["ACLR_L"], stat["ACLR_R"] = np.mean([left]), np.mean([right])
    stat["ACLR_AVG"] = (stat["ACLR_L"] + stat["ACLR_R"]) / 2
    return stat


def calculate_metrics_many(args, stats, predictions, ground_truths):
    """`calculate_metrics` for K runs at once (lockstep sweeps, opendpd_amd/sweep.py): the K prediction arrays (segments, nperseg, 2) of one
    split are stacked and every FFT / Welch pass is ONE call over (K x segments) rows; a spectrum is computed row by row, so each run's
    numbers are those of its own `calculate_metrics` call bit for bit (tests/test_metrics_data_cpu.py).  Runs that share their ground truth
    (the same dataset split: the usual case) share its spectrum."""
    K = len(predictions)
    if K == 0:
        return stats
    shapes = {p.shape for p in predictions} | {g.shape for g in ground_truths}
    if len(shapes) != 1 or predictions[0].ndim != 3:
        return [calculate_metrics(args, st, p, g) for st, p, g in zip(stats, predictions, ground_truths)]
    from scipy.signal import welch
    S, N = predictions[0].shape[:2]
    nperseg, n_sub = args.nperseg, args.n_sub_ch
    P = np.stack(predictions)                                                   # (K, S, N, 2)
    same_truth = all(g is ground_truths[0] for g in ground_truths)
    G = ground_truths[0][None] if same_truth else np.stack(ground_truths)      # (1 | K, S, N, 2)
    # NMSE (metrics.py:42-52)
    err = np.square(G[..., 0] - P[..., 0]) + np.square(G[..., 1] - P[..., 1])
    energy = np.square(G[..., 0]) + np.square(G[..., 1])
    nmse = np.mean(10 * np.log10(np.mean(err, axis=-1) / np.mean(energy, axis=-1)), axis=-1)
    # EVM (metrics.py:55-108)
    sp = np.fft.fftshift(np.fft.fft(IQ_to_complex(P), n=nperseg, axis=-1), axes=-1)
    sg = np.fft.fftshift(np.fft.fft(IQ_to_complex(G), n=nperseg, axis=-1), axes=-1)
    freq = np.fft.fftshift(np.fft.fftfreq(N, d=1 / int(800e6)))
    lo, _, w = _main_channel_bins(freq, args.bw_main_ch, n_sub)
    e = np.zeros((K, S, n_sub))
    for c in range(n_sub):
        sl = slice(lo + c * w, lo + (c + 1) * w)
        e[:, :, c] = np.mean(np.abs(sp[:, :, sl] - sg[:, :, sl]), axis=-1)
        e[:, :, c] = e[:, :, c] / np.mean(np.abs(sg[:, :, sl]), axis=-1)
    evm = 20 * np.log10(np.mean(e.mean(axis=-1), axis=-1))
    # ACLR (metrics.py:111-187): one Welch call over all K x S rows
    f2, ps = welch(IQ_to_complex(P).reshape(K * S, N), fs=args.input_signal_fs, nperseg=nperseg, return_onesided=False, scaling="spectrum", axis=-1)
    half = int(nperseg / 2)
    f2 = np.concatenate((f2[half:], f2[:half]))
    ps = np.concatenate((ps[..., half:], ps[..., :half]), axis=-1).reshape(K, S, -1)
    lo2, hi2, w2 = _main_channel_bins(f2, args.bw_main_ch, n_sub)
    for k, st in enumerate(stats):
        psd = np.mean(ps[k], axis=0)
        sub = np.zeros(n_sub)
        for c in range(n_sub):
            sub[c] = np.sum(psd[lo2 + c * w2:lo2 + (c + 1) * w2])
        ref = sub.max()
        left = np.mean(10 * np.log10(np.sum(psd[lo2 - w2:lo2]) / ref))
        right = np.mean(10 * np.log10(np.sum(psd[hi2:hi2 + w2]) / ref))
        st["NMSE"], st["EVM"] = nmse[k], evm[k]
        st["ACLR_L"], st["ACLR_R"] = np.mean([left]), np.mean([right])
        st["ACLR_AVG"] = (st["ACLR_L"] + st["ACLR_R"]) / 2
    return stats
