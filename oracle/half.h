/* TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
 *
 * Software IEEE-754 binary16 <-> binary32 conversion (round-to-nearest-even),
 * needed because gcc 11 on x86-64 has no _Float16.  The reference stores the
 * correlation volume and lookup result as c10::Half; c10::Half arithmetic is
 * "convert to float, operate in float, round back to half"
 * (/root/reference/src/correlation_kernels.cu:55-65 uses `s * scalar_t(w)` and
 * `corr += ...` on c10::Half), which these helpers reproduce bit-for-bit.
 */
#ifndef DBA_ORACLE_HALF_H
#define DBA_ORACLE_HALF_H
#include <stdint.h>
#include <string.h>

static inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 255) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
  }
  int e = (int)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (e <= 0) {                                   /* subnormal or zero */
    if (e < -10) return sign;                     /* too small: rounds to 0 */
    man |= 0x800000u;                             /* implicit 1 */
    int shift = 14 - e;                           /* 14..24 */
    uint32_t hm = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (hm & 1u))) hm++;
    return (uint16_t)(sign | hm); /* may carry into exponent: correct */
  }
  uint32_t hm = man >> 13;
  uint32_t rem = man & 0x1fffu;
  uint16_t out = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) out++; /* carry ok */
  return out;
}

#endif
