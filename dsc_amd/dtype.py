"""dtypes of the DSC operator surface.

The four enum values are ABI (dsc/include/dsc_dtype.h:51-56) and the names `Dtype`, `ScalarType`, `NP_TO_DTYPE`,
`DTYPE_TO_NP`, `DTYPE_SIZE`, `TYPENAME_LOOKUP` and `DTYPE_CONVERSION_TABLES` are what `dsc_amd.tensor` (like the
reference's wrapper) looks up; everything is derived from one table."""
from enum import IntEnum
from typing import Union

import numpy as np

ScalarType = Union[int, float, complex]

# (name, ABI code, numpy dtype, bytes per element, complex?)
_SPEC = (
    ('f32', 0, np.float32, 4, False),
    ('f64', 1, np.float64, 8, False),
    ('c32', 2, np.complex64, 8, True),
    ('c64', 3, np.complex128, 16, True),
)


class Dtype(IntEnum):
    F32, F64, C32, C64 = (code for _, code, _, _, _ in _SPEC)

    def __str__(self) -> str:
        return _SPEC[int(self)][0]

    __repr__ = __str__

    @staticmethod
    def is_complex(x: 'Dtype') -> bool:
        return _SPEC[int(x)][4]


TYPENAME_LOOKUP = {Dtype(code): name for name, code, _, _, _ in _SPEC}
DTYPE_SIZE = {Dtype(code): size for _, code, _, size, _ in _SPEC}
NP_TO_DTYPE = {np.dtype(npt): Dtype(code) for _, code, npt, _, _ in _SPEC}
DTYPE_TO_NP = {v: k for k, v in NP_TO_DTYPE.items()}


def _promote(a: Dtype, b: Dtype) -> Dtype:
    """dsc/include/dsc_dtype.h:73-78: complex wins over real, then the wider of the two — except that a real f64 does NOT
    widen a c32 (F64 x C32 -> C32)."""
    if Dtype.is_complex(a) != Dtype.is_complex(b):
        return a if Dtype.is_complex(a) else b
    return a if int(a) >= int(b) else b


DTYPE_CONVERSION_TABLES = [[_promote(a, b) for b in Dtype] for a in Dtype]
