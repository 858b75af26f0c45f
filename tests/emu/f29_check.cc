// TEST INFRASTRUCTURE: drives the F29 field primitives (zk_field29.h, the host path = the device source) on operands read
// from stdin so that tests/test_f29.py can compare them with Python integers at the extreme limb bounds the bound
// discipline allows.  Line format:  <field> <op> <L limbs of a> <L limbs of b>   ->   L limbs of the result.
#include <initializer_list>
#include <stdio.h>
#include <string.h>

#include "zk_curve29.h"
using namespace zk;

// Fq2 operations: operands are (a0 a1) (b0 b1), the result is (c0 c1)
template <class P>
static bool run_x2(const char* op) {
    constexpr int L = F29<P>::L;
    if (strncmp(op, "x2", 2)) return false;
    Fe29x2<P> a, b, r;
    for (Fe29<P>* c : {&a.c0, &a.c1, &b.c0, &b.c1})
        for (int i = 0; i < L; i++) scanf("%u", &c->v[i]);
    if (!strcmp(op, "x2mul4k1")) fe29_mul(r, a, b, F29<P>::BIAS4K1);
    else if (!strcmp(op, "x2mul8k2")) fe29_mul(r, a, b, F29<P>::BIAS8K2);
    else if (!strcmp(op, "x2mul16k2")) fe29_mul(r, a, b, F29<P>::BIAS16K2);
    else if (!strcmp(op, "x2sqr8k2")) fe29_sqr(r, a, F29<P>::BIAS8K2);
    else if (!strcmp(op, "x2sqr16k2")) fe29_sqr(r, a, F29<P>::BIAS16K2);
    else if (!strcmp(op, "x2refresh")) fe29_refresh(r, a);
    else if (!strcmp(op, "x2iszero")) {
        printf("%d\n", (int)fe29_is_zero_mod_p(a, b.c0.v[0], b.c0.v[1]));
        return true;
    } else {
        printf("bad op\n");
        return true;
    }
    for (int i = 0; i < L; i++) printf("%u ", r.c0.v[i]);
    for (int i = 0; i < L; i++) printf("%u ", r.c1.v[i]);
    printf("\n");
    return true;
}

template <class P>
static void run(const char* op) {
    constexpr int L = F29<P>::L;
    if (run_x2<P>(op)) return;
    Fe29<P> a, b, r;
    for (int i = 0; i < L; i++) scanf("%u", &a.v[i]);
    for (int i = 0; i < L; i++) scanf("%u", &b.v[i]);
    if (!strcmp(op, "mul")) fe29_mul(r, a, b);
    else if (!strcmp(op, "sqr")) fe29_sqr(r, a);
    else if (!strcmp(op, "mulacc")) {   // a b + c d with c = b reversed, d = a reversed (distinct operands from two vectors)
        Fe29<P> c, d;
        for (int i = 0; i < L; i++) c.v[i] = b.v[i] >> 1, d.v[i] = a.v[i] >> 1;
        fe29_mulacc(r, a, b, c, d);
    }
    else if (!strcmp(op, "sub4k1")) fe29_sub(r, a, b, F29<P>::BIAS4K1);
    else if (!strcmp(op, "sub16k2")) fe29_sub(r, a, b, F29<P>::BIAS16K2);
    else if (!strcmp(op, "sub3")) fe29_sub3(r, a, b, b);
    else if (!strcmp(op, "sub2x")) fe29_sub2x(r, a, b);
    else if (!strcmp(op, "norm")) fe29_norm(r, a);
    else if (!strcmp(op, "carry")) fe29_carry(r, a);
    else if (!strcmp(op, "canon")) fe29_canon(r, a);
    else if (!strcmp(op, "tostd")) {
        Fe<P> s;
        fe29_to_std(s, a);
        for (int i = 0; i < P::N; i++) printf("%u ", s.v[i]);
        printf("\n");
        return;
    } else if (!strcmp(op, "fromstd")) {
        Fe<P> s;
        for (int i = 0; i < P::N; i++) s.v[i] = a.v[i];
        fe29_from_std(r, s);
    } else if (!strcmp(op, "filter")) {
        uint32_t k = 0;
        bool f = fe29_zero_filter(a, b.v[0], b.v[1], k);
        bool e = f && fe29_is_kp(a, k);
        printf("%d %u %d\n", (int)f, k, (int)e);
        return;
    } else {
        printf("bad op\n");
        return;
    }
    for (int i = 0; i < L; i++) printf("%u ", r.v[i]);
    printf("\n");
}

int main() {
    char field[32], op[32];
    while (scanf("%31s %31s", field, op) == 2) {
#define GO(P)                    \
    if (!strcmp(field, #P)) {    \
        run<P>(op);              \
        continue;                \
    }
        GO(PallasFp) GO(PallasFq) GO(Bn254Fq) GO(Bls381Fq)
        printf("bad field\n");
        return 1;
    }
    return 0;
}
