// The ring GEMM (gemm256.hip) compiled once more with the dynamic tile scheduler in its kernels: gemm256_kernel<.., true>,
// reached through vault_gemm256_launch_dyn / vault_gemm256_grouped_launch_dyn when GemmParams::persist bit 0 is set (data-parallel
// steps: GEMMs that share the CUs with RCCL kernels).  See the R256_DYN note in gemm256.hip.
#define R256_DYN 1
#include "gemm256.hip"
