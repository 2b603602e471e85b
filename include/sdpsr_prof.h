/*
 * sdpsr_prof.h -- measurement entry points of libsdpsr_prof.so.  NOT part of the product ABI
 * (include/sdpsr.h, libsdpsr_hip.so): bench.py's roofline leg and the scripts under tools/ time
 * single kernels of the product library through these.  libsdpsr_prof.so links against
 * libsdpsr_hip.so and works on the same sdpsr_ctx.
 */
#ifndef SDPSR_PROF_H
#define SDPSR_PROF_H

#include "sdpsr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Times `reps` back-to-back launches of one hot kernel on ctx's stream with HIP events, on
   resident synthetic data of order n (n is rounded up to the kernel's tile).  kind:
   0 = square int8 MFMA (one channel), 1 = square fp32 MFMA, 2 = square / Q'AQ fp64 MFMA,
   3 = partition refine of n*n signatures with `aux` distinct classes,
   4 = fused gather+projection+signature pass with r = aux basis vectors,
   5 = the tridiagonalisation's symmetric-product (symv) kernel, one launch per column j = 0..n-2
       (average over the n-1 launches), 6 = one whole tridiagonalisation of order n,
   8 = the one-workgroup Jacobi eigensolver on a random symmetric matrix of order n <= 128,
   9 = the label product Y = A(v) W of the module-compression driver (n x n labels with d classes,
       W n x w, G elements per pass): aux = w | G << 8 | d << 12; with bit 30 of aux set the call
       returns in ms_per_launch[0] the largest absolute deviation of sampled rows of Y from a host
       evaluation in extended precision instead of the time.
   10 = the HOST eigensolver of the compressed problem (order n <= 512, one host core, no device work);
   11 = kind 3 as the FIRST refinement of a call meets it: the class-count prediction is reset before every
       repetition (a many-classes input pays the overflowing first pass, the sample and the path it selects);
   ms_per_launch[0] = average milliseconds per launch. */
int sdpsr_profile_kernel(sdpsr_ctx* ctx, int kind, int64_t n, int64_t aux, int reps,
                         double* ms_per_launch);
/* The same measurement with the shader clock sampled meanwhile by a one-wave kernel on a side
   stream (clock64 against the 100 MHz wall_clock64, ~20 us intervals): out[0] = ms per launch,
   out[1] = median shader clock in MHz while the timed launches ran, out[2] = intervals used.
   The int8 squares run power-limited (the clock drops under the kernel); the roofline of
   bench.py reports the fraction of the peak both at the nominal and at this measured clock. */
int sdpsr_profile_clock(sdpsr_ctx* ctx, int kind, int64_t n, int64_t aux, int reps, double* out);
/* Host waits for a stream of ctx since its creation (every wait of the library goes through one function): the host round
   trips of a reduction = the difference around it. */
int sdpsr_profile_host_waits(sdpsr_ctx* ctx, uint64_t* out);
/* Stage 2 of a two-stage tridiagonalisation (band -> tridiagonal by Householder bulge chasing, one workgroup per sweep,
   hand-offs through progress words), built to be measured: A_host n x n dense symmetric with bandwidth b (16, 32 or 64),
   d_host (n), e_host (n - 1) the tridiagonal result, out[0] = kernel milliseconds, out[1] = 1 if the chase gave up. */
int sdpsr_profile_band_chase(sdpsr_ctx* ctx, int64_t n, int b, const double* A_host, double* d_host, double* e_host, double* out);
/* Stage 1 of the two-stage form, measured with library kernels (rocSOLVER panel QR + rocBLAS level-3 updates): dense
   symmetric A (n x n, column-major, host, n a multiple of b) -> band of width b, in place (lower triangle);
   out[0] = ms of the stage (second of two runs), out[1] = ms of its panel factorisations. */
int sdpsr_profile_band_reduce(sdpsr_ctx* ctx, int64_t n, int b, double* A_host, double* out);
/* What the hipGraph cache of the tridiagonalisation (one graph per problem shape and buffer set, kept in ctx) has done
   so far: out[0] = replays of a cached graph, out[1] = misses (a graph of ~2 n nodes built and instantiated on the
   host), out[2] = milliseconds spent building.  A caller that alternates between a few orders pays the build once per
   order. */
int sdpsr_profile_sytrd_graphs(sdpsr_ctx* ctx, double* out);

#ifdef __cplusplus
}
#endif
#endif /* SDPSR_PROF_H */
