/* Diagnostic entry points of libnvq_debug.so (built by `NVQ_DEBUG_TOOLS=1 bash build.sh` with -DNVQ_DEBUG_TOOLS; the shipped
 * libnvq.so does not contain them and has no mutable state).  Used by tools/kernel_phases.py and tools/occupancy_probe.py. */
#ifndef NVQ_DEBUG_H
#define NVQ_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* 0 = normal; 1 = the bf16 conv kernels skip the MFMA section; 2 = they skip the per-chunk global loads after the first
 * chunk.  Results are wrong in modes 1 and 2.  Process-wide. */
int nvq_debug_set_conv_mode(int mode);
/* resident workgroups per CU of conv<2,3,8>, conv<2,3,8,split>, conv<4,3,4>, rdb_tail, wgrad<3,64>, conv<2,3,4> */
int nvq_debug_conv_occupancy(int* out6);
#ifdef __cplusplus
}
#endif
#endif
