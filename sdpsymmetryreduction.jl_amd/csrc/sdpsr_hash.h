// Counter-based random values per class and the 64-bit signature mix.
// Usable from host and device code (plain integer arithmetic).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SDPSR_HD __host__ __device__ __forceinline__
#else
#define SDPSR_HD inline
#endif

SDPSR_HD uint64_t sdpsr_fmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// key of one draw: seed and stream id are folded once on the host.
SDPSR_HD uint64_t sdpsr_stream_key(uint64_t seed, uint64_t stream) {
    return sdpsr_fmix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
}

// 64 random bits of class `label` (1-based) in draw `key`.
SDPSR_HD uint64_t sdpsr_class_bits(uint64_t key, uint32_t label) {
    return sdpsr_fmix64(key + 0x9E3779B97F4A7C15ULL * (uint64_t)label);
}

// uniform [0,1) double, rand(Float64) of src/abstract_part.jl:108
SDPSR_HD double sdpsr_class_uniform(uint64_t key, uint32_t label) {
    return (double)(sdpsr_class_bits(key, label) >> 11) * (1.0 / 9007199254740992.0);
}

// channel t (0..7) int8 value of the class: byte t of the 64 bits, as signed.
SDPSR_HD int sdpsr_class_i8(uint64_t bits, int t) { return (int)(int8_t)(bits >> (8 * t)); }

// integer value in [-vmax, vmax] for the fp32-exact mode (7 bits per channel used).
SDPSR_HD int sdpsr_class_small(uint64_t bits, int t, int vmax) {
    uint32_t b = (uint32_t)(bits >> (8 * t)) & 0xFFu;
    return (int)((b * (uint32_t)(2 * vmax + 1)) >> 8) - vmax;
}

// signature chaining: h' = mix(h, v)
SDPSR_HD uint64_t sdpsr_sig_mix(uint64_t h, uint64_t v) {
    return sdpsr_fmix64(h + 0x9E3779B97F4A7C15ULL + v * 0xD6E8FEB86659FD93ULL);
}

// start of a signature chain from the old label.  Not mixed by itself: every chain goes through
// at least one sdpsr_sig_mix, whose finaliser does the mixing (one 64 x 32 multiply here instead
// of two 64 x 64 ones; 32/64-bit integer multiplies are quarter rate on CDNA and the signature
// kernels are bound by them, not by HBM).
SDPSR_HD uint64_t sdpsr_sig_start(uint32_t label) {
    return 0x51ED270B0F3A4C27ULL + (uint64_t)label * 0xC2B2AE3D27D4EB4FULL;
}

// _clamp_round! / unsafe_round (src/utils.jl:34-53): |a| < atol -> +0.0; else the mantissa in
// [0.5,1) is kept to |scale| = 10^sigdigits steps.  The sign of `scale` carries the rounding rule
// (one wave-uniform select, no extra kernel argument anywhere): scale > 0 rounds the scaled mantissa
// to nearest (the library's default, see sdpsr.h at sdpsr_clamp_round), scale < 0 truncates it like
// the reference's unsafe_trunc(Int, scale * x) / scale (sdpsr_opts.round_mode = SDPSR_ROUND_TRUNC).
SDPSR_HD double sdpsr_clamp_round(double a, double atol, double scale) {
    double aa = a < 0 ? -a : a;
    if (aa < atol) return 0.0;
    const bool trunc_mode = scale < 0;
    const double sc = trunc_mode ? -scale : scale;
    int e;
#if defined(__HIP_DEVICE_COMPILE__)
    double x = frexp(a, &e);
    const double t = sc * x;
    double y = (trunc_mode ? trunc(t) : rint(t)) / sc;
    return ldexp(y, e);
#else
    double x = __builtin_frexp(a, &e);
    const double t = sc * x;
    double y = (trunc_mode ? __builtin_trunc(t) : __builtin_rint(t)) / sc;
    return __builtin_ldexp(y, e);
#endif
}

// The same rule as an injective 64-bit CODE of the rounded value, for signatures: 0 for |a| < atol,
// else (k, e) with k = the rounded / truncated scaled mantissa (|k| in [|scale| / 2, |scale|)) and e the
// binary exponent, packed as a 52-bit and a 12-bit two's complement field.
// Two inputs get the same code exactly when sdpsr_clamp_round gives them the same double (k / scale
// is strictly monotone in k; a mantissa that rounds up to 1.0 is folded onto 0.5 * 2^(e+1), the same
// double) -- so classes formed from the codes are the classes of the rounded values, without the
// fp64 division and the ldexp per entry that the value itself costs.
SDPSR_HD uint64_t sdpsr_round_key(double a, double atol, double scale) {
    double aa = a < 0 ? -a : a;
    if (aa < atol) return 0ull;
    const bool trunc_mode = scale < 0;
    const double sc = trunc_mode ? -scale : scale;
    int e;
#if defined(__HIP_DEVICE_COMPILE__)
    const double x = frexp(a, &e);
    const double t = sc * x;
    double kd = trunc_mode ? trunc(t) : rint(t);
#else
    const double x = __builtin_frexp(a, &e);
    const double t = sc * x;
    double kd = trunc_mode ? __builtin_trunc(t) : __builtin_rint(t);
#endif
    if (kd == sc || kd == -sc) {  // mantissa rounded up to 1.0: the same double as 0.5 * 2^(e+1)
        kd *= 0.5;
        e += 1;
    }
    // |k| <= 1e15 < 2^51 for every sigdigits <= 15; e in [-1074, 1025]: 52 + 12 bits, two's complement fields
    return ((uint64_t)(int64_t)kd & ((1ull << 52) - 1)) | ((uint64_t)((uint32_t)e & 0xFFFu) << 52);
}

