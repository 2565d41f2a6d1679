// Event-profiling classes (see prof.cpp).  Kernel classes bracket exactly one kernel launch; phase classes
// bracket a run of launches inside the engine forward.
#pragma once
#include <hip/hip_runtime.h>

#define ROITR_PROF_CLASSES 19
enum {
    ROITR_PROF_FPS = 0,        // fps_kernel, one launch
    ROITR_PROF_KNN = 1,        // knn_grid_kernel / knn_brute_kernel (the query kernel incl. fused PPF), one launch
    ROITR_PROF_GRID = 2,       // grid_build_kernel
    ROITR_PROF_REPLAY = 3,     // knn_replay_kernel
    ROITR_PROF_PH_GEOM = 4,    // engine phases
    ROITR_PROF_PH_ENC = 5,
    ROITR_PROF_PH_GEO = 6,
    ROITR_PROF_PH_DEC = 7,
    ROITR_PROF_PH_MATCH = 8,
    ROITR_PROF_PH_FORWARD = 9,
    ROITR_PROF_OT = 10,        // ot_kernel
    ROITR_PROF_LOCAL_ATTN = 11,// local_attn_kernel
    ROITR_PROF_GEMM = 12,      // gemm_kernel: the "bytes" field carries FLOPs (2*M*N*K*batch)
    ROITR_PROF_MHA = 13,       // mha_kernel
    ROITR_PROF_GEO_EMBED = 14, // geo_embed_kernel (GEMM form): "bytes" carries FLOPs
    ROITR_PROF_GEO_TABLE = 15, // geo_table_kernel: HBM bytes
    ROITR_PROF_GEO_ALGO = 16,  // no time: the FLOPs the GEMM form of the embedding would have spent on the rows geo_table_kernel served
    ROITR_PROF_LOCAL_BLOCK = 17,// local_block_kernel: "bytes" = HBM bytes, aux = the FLOPs of its three on-chip GEMMs
    ROITR_PROF_GEMM_HBM = 18   // gemm_kernel launches whose roof is HBM: algorithmic FLOPs per algorithmic byte below the machine balance
                               // (roitr_gemm_prof_class); same fields as ROITR_PROF_GEMM, which keeps the MFMA-roofed launches
};

void roitr_prof_begin(int cls, double bytes, hipStream_t st);
void roitr_prof_begin2(int cls, double bytes, double aux, hipStream_t st);   // aux: second accumulator (prof.cpp roitr_prof_read_aux)
// a launch over a batch list whose live length is a device int: bytes / aux per unit, multiplied by *dev_units when the bracket is folded
void roitr_prof_begin_live(int cls, double bytes_per_unit, double aux_per_unit, const int* dev_units, hipStream_t st);
void roitr_prof_end(int cls, hipStream_t st);
void roitr_prof_note(int cls, double v);   // adds v to the class total without a timed bracket
extern "C" void roitr_prof_enable(int on);
extern "C" int roitr_prof_is_enabled(void);
extern "C" void roitr_prof_reset(void);
extern "C" void roitr_prof_next_bytes(int cls, double bytes);
extern "C" int roitr_prof_read(int cls, double* ms, long* launches, double* bytes);
extern "C" int roitr_prof_read_aux(int cls, double* aux);

// Algorithmic HBM bytes of one GEMM launch: every operand element read once, every result written once (rows gathered from a
// larger tensor count as read once each, a ragged batch is priced at its bounding M x N).  bench.py `roofline` of gemm_kernel.
struct RoitrGemm;
double roitr_gemm_algorithmic_bytes(const RoitrGemm* g);
// ROITR_PROF_GEMM (MFMA-roofed) or ROITR_PROF_GEMM_HBM: 2 M N K over the algorithmic bytes against peak FLOP/s over peak HBM bytes/s of
// the operand dtype (fp32: 157.3 T / 8 T = 19.7 FLOP per byte; bf16 operands: 2500 / 8 = 312)
int roitr_gemm_prof_class(const RoitrGemm* g);
