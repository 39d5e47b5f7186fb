/* libm_f32.h — atanf / atan2f / acosf in float, restated so that the CPU oracle and the HIP kernels return the same bits.
 *
 * TEST INFRASTRUCTURE (part of oracle/, see ope_oracle.h).  PCL's computePairFeatures calls acos / atan2 on floats
 * (uPCL features/src/pfh.cpp); which bits come back is the C library's business, and the GPU's math library rounds a few
 * results per million differently from glibc — enough to move a pair feature across an FPFH bin edge.  These are the
 * classic fdlibm float algorithms (Sun Microsystems' freely distributable routines as glibc's flt-32 directory carries
 * them): IEEE float operations in a fixed order, nothing machine-specific.  tests/test_oracle_kat.py checks them against
 * the C library of the box bit for bit (glibc 2.35: 0 differences in 2e7 atan2f, 3e8 acosf arguments), so the oracle's
 * results are what they were with libm; object-pose-estimation_amd/csrc/libm_f32.hpp is the same text for the device. */
#ifndef OPE_ORACLE_LIBM_F32_H
#define OPE_ORACLE_LIBM_F32_H
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline uint32_t lmf_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float lmf_from(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

static inline float lmf_atanf(float x) {
  static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                               6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const float one = 1.0f, huge = 1.0e30f;
  float w, s1, s2, z;
  int32_t ix, hx, id;
  hx = (int32_t)lmf_bits(x);
  ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) { /* |x| >= 2^25 */
    if (ix > 0x7f800000) return x + x; /* NaN */
    if (hx > 0) return atanhi[3] + atanlo[3];
    else return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {    /* |x| < 0.4375 */
    if (ix < 0x31000000) {  /* |x| < 2^-29 */
      if (huge + x > one) return x;
    }
    id = -1;
  } else {
    x = lmf_from((uint32_t)ix);   /* fabsf */
    if (ix < 0x3f980000) {        /* |x| < 1.1875 */
      if (ix < 0x3f300000) { id = 0; x = ((float)2.0 * x - one) / ((float)2.0 + x); }
      else { id = 1; x = (x - one) / (x + one); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - (float)1.5) / (one + (float)1.5 * x); }
      else { id = 3; x = -(float)1.0 / x; }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}

static inline float lmf_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  float z;
  int32_t k, m, hx, hy, ix, iy;
  hx = (int32_t)lmf_bits(x); ix = hx & 0x7fffffff;
  hy = (int32_t)lmf_bits(y); iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   /* NaN */
  if (hx == 0x3f800000) return lmf_atanf(y);              /* x = 1.0 */
  m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                /* 2*sign(x) + sign(y) */
  if (iy == 0) {
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      case 3: return -pi - tiny;
    }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return (float)3.0 * pi_o_4 + tiny;
        case 3: return (float)-3.0 * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return zero;
        case 1: return -zero;
        case 2: return pi + tiny;
        case 3: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  k = (iy - ix) >> 23;
  if (k > 60) z = pi_o_2 + (float)0.5 * pi_lo;       /* |y/x| > 2^60 */
  else if (hx < 0 && k < -60) z = 0.0f;              /* |y|/x < -2^60 */
  else { float q = y / x; z = lmf_atanf(lmf_from(lmf_bits(q) & 0x7fffffffu)); }
  switch (m) {
    case 0: return z;
    case 1: return lmf_from(lmf_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

static inline float lmf_sqrtf(float x) { return sqrtf(x); }   /* IEEE square root */

static inline float lmf_acosf(float x) {
  const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
              pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  float z, p, q, r, w, s, c, df;
  int32_t hx, ix;
  hx = (int32_t)lmf_bits(x);
  ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) {            /* |x| == 1 */
    if (hx > 0) return 0.0f;
    else return pi + (float)2.0 * pio2_lo;
  } else if (ix > 0x3f800000) {
    return (x - x) / (x - x);        /* NaN */
  }
  if (ix < 0x3f000000) {             /* |x| < 0.5 */
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;   /* |x| < 2^-57 */
    z = x * x;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (hx < 0) {               /* x < -0.5 */
    z = (one + x) * (float)0.5;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    s = lmf_sqrtf(z);
    r = p / q;
    w = r * s - pio2_lo;
    return pi - (float)2.0 * (s + w);
  } else {                           /* x > 0.5 */
    z = (one - x) * (float)0.5;
    s = lmf_sqrtf(z);
    df = lmf_from(lmf_bits(s) & 0xfffff000u);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return (float)2.0 * (df + w);
  }
}

/* cos and sin of an angle in [0, pi/3] — all eigen33 asks for (theta = atan2(...) / 3): the Taylor series in double (terms to
 * x^22 / x^23: below 1e-21 at pi/3), rounded to float once.  IEEE double operations in a fixed order: the same bits anywhere.
 * (The C library's cosf / sinf of this box are a different, equally valid rounding of the same values in a few arguments per
 * thousand; PCL's Scalar = float eigen33 gets whatever its platform's libm returns.) */
static inline void lmf_cos_sin_small(float xf, float *c, float *s) {
  const double x = (double)xf, z = x * x;
  const double pc = 1.0 + z * (-1.0 / 2.0 + z * (1.0 / 24.0 + z * (-1.0 / 720.0 + z * (1.0 / 40320.0 + z * (-1.0 / 3628800.0 + z * (1.0 / 479001600.0 +
                    z * (-1.0 / 87178291200.0 + z * (1.0 / 20922789888000.0 + z * (-1.0 / 6402373705728000.0 + z * (1.0 / 2432902008176640000.0 +
                    z * (-1.0 / 1124000727777607680000.0)))))))))));
  const double ps = 1.0 + z * (-1.0 / 6.0 + z * (1.0 / 120.0 + z * (-1.0 / 5040.0 + z * (1.0 / 362880.0 + z * (-1.0 / 39916800.0 + z * (1.0 / 6227020800.0 +
                    z * (-1.0 / 1307674368000.0 + z * (1.0 / 355687428096000.0 + z * (-1.0 / 121645100408832000.0 + z * (1.0 / 51090942171709440000.0 +
                    z * (-1.0 / 25852016738884976640000.0)))))))))));
  *c = (float)pc;
  *s = (float)(x * ps);
}

#endif
