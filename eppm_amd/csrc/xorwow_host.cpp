// xorwow_host.cpp -- host side of the XORWOW generator the reference draws from through cuRAND
// (curand_init(1234, block_id, 0) / curand(), bao_pmflow_kernel.cu:68,94-95,1546-1547).
//
// cuRAND is not vendored in the reference.  This is the published algorithm: Marsaglia's xorwow
// ("Xorshift RNGs", JSS 8(14), 2003) with cuRAND's seeding -- seed scrambling, then subsequence n
// starts n * 2^67 draws into the stream.  The skip is a multiplication by the 160x160 GF(2) matrix of
// the xorshift recurrence raised to the required power; the Weyl counter advances arithmetically.
#include <mutex>
#include <string.h>
#include <vector>

#include "eppm_internal.h"

namespace eppm {

namespace {

struct Mat { uint32_t r[160][5]; };

inline void step_v(uint32_t v[5])
{
    const uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

void vec_mat(const uint32_t v[5], const Mat& m, uint32_t out[5])
{
    uint32_t a[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 160; i++)
        if (v[i >> 5] & (1u << (i & 31)))
            for (int k = 0; k < 5; k++) a[k] ^= m.r[i][k];
    memcpy(out, a, sizeof(a));
}

void mat_mul(const Mat& a, const Mat& b, Mat& out)
{
    Mat t;
    for (int i = 0; i < 160; i++) vec_mat(a.r[i], b, t.r[i]);
    out = t;
}

void mat_identity(Mat& m)
{
    memset(&m, 0, sizeof(m));
    for (int i = 0; i < 160; i++) m.r[i][i >> 5] = 1u << (i & 31);
}

struct Tables {
    std::vector<Mat> one;   // one[k] = M^(2^k), k = 0..63
    std::vector<Mat> seq;   // seq[k] = M^(2^(67+k)), k = 0..39
};

const Tables& tables()
{
    static Tables t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.one.resize(64);
        t.seq.resize(40);
        for (int i = 0; i < 160; i++) {
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[i >> 5] = 1u << (i & 31);
            step_v(v);
            memcpy(t.one[0].r[i], v, sizeof(v));
        }
        for (int k = 1; k < 64; k++) mat_mul(t.one[k - 1], t.one[k - 1], t.one[k]);
        Mat cur;
        mat_mul(t.one[63], t.one[63], cur);                   // 2^64
        for (int k = 64; k < 67; k++) mat_mul(cur, cur, cur);  // 2^67
        t.seq[0] = cur;
        for (int k = 1; k < 40; k++) mat_mul(t.seq[k - 1], t.seq[k - 1], t.seq[k]);
    });
    return t;
}

}  // namespace

void xorwow_init(XorwowState* s, unsigned long long seed, unsigned long long subsequence)
{
    const Tables& t = tables();
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
    for (int k = 0; k < 40; k++)
        if (subsequence & (1ULL << k)) vec_mat(s->v, t.seq[k], s->v);
}

uint32_t xorwow_next(XorwowState* s)
{
    step_v(s->v);
    s->d += 362437u;
    return s->v[4] + s->d;
}

void xorwow_skip_matrix(unsigned long long n, uint32_t* out)
{
    const Tables& t = tables();
    Mat acc;
    mat_identity(acc);
    for (int k = 0; k < 64; k++)
        if (n & (1ULL << k)) mat_mul(acc, t.one[k], acc);
    memcpy(out, acc.r, sizeof(acc.r));
}

}  // namespace eppm
