// common.h -- shared constants and small helpers for the gfx950 (CDNA4) kernels of libbohip.
// Everything here is FP64: the acceptance bar is 1e-6 relative on sigma^2 = s_f^2 - v'v, which
// cancels catastrophically near observations, so no reduced-precision MFMA is usable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bohip {

// ---- tiling of every dense FP64 contraction --------------------------------------------------
// Workgroup tile 128 rows x 64 columns, 4 waves (2 x 2), wave tile 64 x 32 = 8 x 4 MFMA groups of 8 x 8.
// K is consumed in chunks of KC = 16 doubles (128 B = one cache line per row).  See gemm_core.h.
constexpr int TILE = 128;      // row-tile height and the block size of all blocked algorithms
constexpr int CTILE = 64;      // column-tile width of the contraction engine
constexpr int KC = 16;
constexpr int GEMM_THREADS = 256;

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

// bound of every in-kernel wait of the dataflow factorisation, in ticks of wall_clock64() (100 MHz): 200 ms (kernels_chol.hip)
constexpr unsigned long long CH_SPIN_TICKS_DEFAULT = 20000000ull;

// INLINE-ASM 16-BYTE STORES carry "s_nop 1" behind them.  gfx9 has a hazard: a VMEM store of more than 64 bits of data followed by
// a VALU instruction that writes one of the store's data VGPRs needs a wait state.  hipcc inserts it for stores it emitted
// itself, but its hazard recogniser does not look inside an inline-asm string, so `global_store_dwordx4 ... sc1` written as asm
// (the atomic builtins stop at 8 bytes) followed by ordinary code that reuses the data registers stored a corrupted low dword:
// measured in the inverter (kernels_chol.hip) as the first double of every other 16-byte piece off by ~1e-7 relative.

struct Best {
    double val;
    long long idx;
};

}  // namespace bohip
