/* tools/szn_debug.h -- debug aid for tools/contention.py, kept out of the product header include/szn.h.
 * `blocks` workgroups x 256 threads spinning for `cycles` shader clocks on `stream`; blocks < 0: |blocks| workgroups of a
 * ~100-VGPR variant that cannot share a SIMD with the persistent conv kernels.  sink: any 4 device bytes. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
int szn_debug_spin(int blocks, long long cycles, void* sink, void* stream);
#ifdef __cplusplus
}
#endif
