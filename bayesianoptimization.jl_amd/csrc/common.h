// common.h -- shared constants and small helpers for the gfx950 (CDNA4) kernels of libbohip.
// Everything here is FP64: the acceptance bar is 1e-6 relative on sigma^2 = s_f^2 - v'v, which
// cancels catastrophically near observations, so no reduced-precision MFMA is usable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bohip {

// ---- tiling of every dense FP64 contraction --------------------------------------------------
// Workgroup tile 128 x 128, 4 waves (2 x 2), each wave 64 x 64 = 8 x 8 MFMA groups of 8 x 8.
// K is consumed in chunks of KC = 16 doubles (128 B = one cache line per row).
// LDS row stride 17 doubles (136 B): conflict-free for the ds_read_b64 fragment pattern (bank pair
// (34*row + 2*col) mod 64 is distinct over the 8 rows x 2 columns a 32-lane half touches) and small
// enough that THREE 128x64 workgroups (52.2 KB each) fit the 160 KB LDS of a CU.
constexpr int TILE = 128;
constexpr int KC = 16;
constexpr int LDSROW = KC + 1;
constexpr int TILE_LDS_DOUBLES = TILE * LDSROW;  // one operand tile, one buffer
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_BYTES = 2 /*buffers*/ * 2 /*A,B*/ * TILE_LDS_DOUBLES * 8;  // 69632 B -> 2 WGs/CU

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// kernel ids / acquisition ids mirror include/bohip.h
enum { KERN_SEARD = 0, KERN_SEISO = 1, KERN_MAT52ARD = 2 };
enum { ACQ_EI = 0, ACQ_PI = 1, ACQ_UCB = 2, ACQ_MI = 3, ACQ_MAXMEAN = 4 };

constexpr int DMAX = 64;  // largest supported input dimension (hyper-parameters live in kernel args)

struct KernelHyper {
    int kern;
    int d;
    double sigma2;     // exp(2 logsig)
    double il2[DMAX];  // exp(-2 loglen_k)
};

struct Best {
    double val;
    long long idx;
};

}  // namespace bohip
