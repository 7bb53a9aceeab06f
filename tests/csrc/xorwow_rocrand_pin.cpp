/* Pins the oracle's XORWOW transition and its 2^67 "subsequence" jump against an INDEPENDENT implementation that ships
 * in the ROCm image: rocRAND's precomputed GF(2) jump matrices (rocrand/rocrand_xorwow_precomputed.h:
 * h_xorwow_jump_matrices[k] = A^(4^k), h_xorwow_sequence_jump_matrices[k] = A^(4^k * 2^67), applied as in
 * rocrand_xorwow.h:51-65,181-208).  rocRAND's seed scrambling constants differ from cuRAND's, the xorshift transition A
 * and the meaning of "subsequence n = n * 2^67 draws ahead" do not; so the matrices are applied to the ORACLE's own
 * seeded states and must reproduce the oracle's states.
 * Build: g++ -O2 -I/opt/rocm/include xorwow_rocrand_pin.cpp <oracle>/libeppm_oracle.so */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define __device__
#include <rocrand/rocrand_xorwow_precomputed.h>

extern "C" {
typedef struct { uint32_t v[5]; uint32_t d; } orc_xorwow;
void orc_xorwow_init(orc_xorwow* s, unsigned long long seed, unsigned long long subsequence);
uint32_t orc_xorwow_next(orc_xorwow* s);
void orc_xorwow_skip(orc_xorwow* s, unsigned long long n);
}

static void mul_mat_vec_inplace(const unsigned int* m, unsigned int* v)      /* rocrand_xorwow.h:51-65 */
{
    unsigned int r[XORWOW_N] = {0};
    for (int ij = 0; ij < XORWOW_N * XORWOW_M; ij++) {
        const int i = ij / XORWOW_M, j = ij % XORWOW_M;
        const unsigned int b = (v[i] & (1U << j)) ? 0xffffffff : 0x0;
        for (int k = 0; k < XORWOW_N; k++) r[k] ^= b & m[i * XORWOW_M * XORWOW_N + j * XORWOW_N + k];
    }
    memcpy(v, r, sizeof(r));
}
static void jump(unsigned long long v, const unsigned int mats[XORWOW_JUMP_MATRICES][XORWOW_SIZE], unsigned int* x)   /* :181-208 */
{
    unsigned int mi = 0;
    while (v > 0) {
        const unsigned int is = (unsigned int)v & ((1 << XORWOW_JUMP_LOG2) - 1);
        for (unsigned int i = 0; i < is; i++) mul_mat_vec_inplace(mats[mi], x);
        mi++;
        v >>= XORWOW_JUMP_LOG2;
    }
}

int main()
{
    int bad = 0;
    const unsigned long long seeds[] = {1234ULL, 0ULL, 1ULL, 0xdeadbeefcafeULL, ~0ULL};
    const unsigned long long subs[] = {1, 2, 3, 5, 111, 1000, 4095, 28000, 1000003, (1ULL << 20) + 7};
    const unsigned long long offs[] = {1, 2, 7, 48, 512, 3024, 3072, 30720, 123456789ULL, (1ULL << 40) + 3};
    for (unsigned long long seed : seeds) {
        orc_xorwow base;
        orc_xorwow_init(&base, seed, 0);
        /* (a) subsequence n: oracle state == rocRAND's A^(n*2^67) applied to the oracle's subsequence-0 state */
        for (unsigned long long n : subs) {
            orc_xorwow o;
            orc_xorwow_init(&o, seed, n);
            unsigned int x[5];
            memcpy(x, base.v, sizeof(x));
            jump(n, h_xorwow_sequence_jump_matrices, x);
            if (memcmp(x, o.v, sizeof(x)) || o.d != base.d) { printf("subsequence mismatch seed %llx n %llu\n", seed, n); bad++; }
        }
        /* (b) offsets: oracle skip (and plain stepping, where short) == rocRAND's A^k */
        for (unsigned long long k : offs) {
            orc_xorwow o = base;
            orc_xorwow_skip(&o, k);
            unsigned int x[5];
            memcpy(x, base.v, sizeof(x));
            jump(k, h_xorwow_jump_matrices, x);
            if (memcmp(x, o.v, sizeof(x)) || o.d != base.d + (uint32_t)k * 362437u) { printf("offset mismatch seed %llx k %llu\n", seed, k); bad++; }
            if (k <= 30720) {
                orc_xorwow st = base;
                for (unsigned long long q = 0; q < k; q++) orc_xorwow_next(&st);
                if (memcmp(x, st.v, sizeof(x)) || st.d != o.d) { printf("stepping mismatch seed %llx k %llu\n", seed, k); bad++; }
            }
        }
    }
    printf(bad ? "FAILED: %d mismatches\n" : "OK: oracle XORWOW transition and 2^67 jumps equal rocRAND's precomputed matrices\n", bad);
    return bad ? 1 : 0;
}
