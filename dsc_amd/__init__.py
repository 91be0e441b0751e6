"""dsc_amd — MI355X (gfx950) backend for the FFT hot path of dspcraft/dsc.

Python mirror of the reference's `dsc` package for the operators on that path
(python/dsc/__init__.py:7-57): init / clear, Tensor, from_numpy, mul, sum / mean / max / min,
fft / ifft / rfft / irfft, plus filter_fft (fused README filterFFT).  Everything calls the
C ABI in include/dsc_mi355x.h through ctypes; importing this package without the built
library raises."""
from .context import clear, init, last_fft_path, shutdown, synchronize, used_mem
from .dtype import Dtype
from .tensor import (Tensor, absolute, add, angle, conj, imag, real, empty, fft, fftfreq, filter_fft, from_numpy, ifft, irfft, max, mean, min, mul, plan_fft, rfft, rfftfreq, sub, sum, transpose,
                     true_div)

from .profiler import profile, start_recording, stop_recording  # noqa: E402

__all__ = ['profile', 'start_recording', 'stop_recording', 'init', 'clear', 'shutdown', 'synchronize', 'used_mem', 'last_fft_path', 'Dtype', 'Tensor', 'empty',
           'from_numpy', 'mul', 'add', 'sub', 'true_div', 'absolute', 'angle', 'conj', 'real', 'imag', 'sum', 'mean', 'max', 'min', 'plan_fft', 'fft', 'ifft', 'rfft', 'irfft', 'filter_fft', 'transpose', 'fftfreq', 'rfftfreq']
