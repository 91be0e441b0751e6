"""dtypes of the DSC operator surface (mirror of the reference's python/dsc/dtype.py:15-65;
enum values are ABI, dsc/include/dsc_dtype.h:51-56)."""
from enum import Enum
from typing import Union

import numpy as np

ScalarType = Union[int, float, complex]


class Dtype(Enum):
    F32 = 0
    F64 = 1
    C32 = 2
    C64 = 3

    def __repr__(self) -> str:
        return TYPENAME_LOOKUP[self]

    def __str__(self) -> str:
        return repr(self)

    @staticmethod
    def is_complex(x: 'Dtype') -> bool:
        return x in (Dtype.C32, Dtype.C64)


TYPENAME_LOOKUP = {Dtype.F32: 'f32', Dtype.F64: 'f64', Dtype.C32: 'c32', Dtype.C64: 'c64'}
DTYPE_SIZE = {Dtype.F32: 4, Dtype.F64: 8, Dtype.C32: 8, Dtype.C64: 16}
NP_TO_DTYPE = {
    np.dtype(np.float32): Dtype.F32,
    np.dtype(np.float64): Dtype.F64,
    np.dtype(np.complex64): Dtype.C32,
    np.dtype(np.complex128): Dtype.C64,
}
DTYPE_TO_NP = {v: k for k, v in NP_TO_DTYPE.items()}
# dsc/include/dsc_dtype.h:73-78 (F64 x C32 -> C32)
DTYPE_CONVERSION_TABLES = [
    [Dtype.F32, Dtype.F64, Dtype.C32, Dtype.C64],
    [Dtype.F64, Dtype.F64, Dtype.C32, Dtype.C64],
    [Dtype.C32, Dtype.C32, Dtype.C32, Dtype.C64],
    [Dtype.C64, Dtype.C64, Dtype.C64, Dtype.C64],
]
