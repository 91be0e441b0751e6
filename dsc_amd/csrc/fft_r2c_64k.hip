// placeholder: replaced by the register-resident kernels
#include "kernels.h"
size_t dsc_r2c64k_table_bytes() { return 0; }
void dsc_r2c64k_build_tables(void *) {}
void dsc_launch_rfft64k(const float *, void *, int, const void *, int, hipStream_t) {}
void dsc_launch_irfft64k(const void *, float *, int, const void *, int, hipStream_t) {}
void dsc_launch_filter64k(const float *, const void *, float *, int, const void *, int, hipStream_t) {}
