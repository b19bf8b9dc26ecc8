/* Test infrastructure: is  q = s r, e = fma(-9, q, s), q' = fma(e, r, q)  with r = RN(1/9)  the IEEE quotient s / 9.0f?
 * (the lean kernel's division, mc-cnn_amd/csrc/cbca_lean.hip: div9 / div9_ok).  Walks every `stride`-th float bit pattern
 * (stride 1: all 2^32) plus the neighbourhoods of the range limits; prints the mismatch counts inside / outside the range
 * 2^-95 <= |s| < 2^125 in which the kernel uses the short form.  gcc -O2 -fopenmp [-mfma] -ffp-contract=off div9_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int check(uint32_t u, unsigned long long *nin, unsigned long long *bad_in, unsigned long long *bad_out)
{
	float s;
	memcpy(&s, &u, 4);
	if (s != s) return 0;
	const float r = 0x1.c71c72p-4f;
	const float q = s * r, e = fmaf(-9.0f, q, s), q2 = fmaf(e, r, q), want = s / 9.0f;
	uint32_t a, w;
	memcpy(&a, &q2, 4); memcpy(&w, &want, 4);
	const int in = fabsf(s) >= 0x1p-95f && fabsf(s) < 0x1p125f;
	*nin += in;
	if (a != w) { if (in) ++*bad_in; else ++*bad_out; }
	return 0;
}

int main(int argc, char **argv)
{
	const long long stride = argc > 1 ? atoll(argv[1]) : 1;
	unsigned long long nin = 0, bad_in = 0, bad_out = 0;
#pragma omp parallel for reduction(+ : nin, bad_in, bad_out) schedule(static)
	for (long long b = 0; b < (1LL << 32); b += stride) check((uint32_t)b, &nin, &bad_in, &bad_out);
	/* the range limits, both signs, +-2^16 patterns around each */
	const float lim[2] = {0x1p-95f, 0x1p125f};
	for (int k = 0; k < 2; ++k) {
		uint32_t u;
		memcpy(&u, &lim[k], 4);
		for (long long t = -65536; t <= 65536; ++t) {
			check((uint32_t)(u + t), &nin, &bad_in, &bad_out);
			check((uint32_t)((u + t) | 0x80000000u), &nin, &bad_in, &bad_out);
		}
	}
	printf("%llu %llu %llu\n", nin, bad_in, bad_out);
	return 0;
}
