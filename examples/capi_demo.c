/* The drop-in boundary from plain C: no C++, no Python, no torch -- only include/enoki_hip.h and libenoki-hip.so.
 *
 *     gcc -std=c11 -Iinclude examples/capi_demo.c -Lenoki_amd -lenoki-hip -Wl,-rpath,$PWD/enoki_amd -lm -o capi_demo
 *
 * Computes y = hsum(fmadd(a, x, b)) and scatter_add(table, fmadd(a, x, b), idx) for n elements on the GPU and checks
 * both against a host loop (integer-valued inputs, so the fp results are exact).  Exit status 0 on success. */
#include <enoki_hip.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != EK_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ek_hip_last_error()); return 1; } \
    } while (0)

int main(void) {
    const size_t n = 1u << 20, k = 4096;
    float *ha = malloc(n * sizeof(float)), *hx = malloc(n * sizeof(float));
    uint32_t *hidx = malloc(n * sizeof(uint32_t));
    double want_sum = 0.0;
    static double want_table[4096];
    for (size_t i = 0; i < n; ++i) {
        ha[i] = (float) (i % 7);
        hx[i] = (float) (i % 5);
        hidx[i] = (uint32_t) ((i * 2654435761u) % k);
        double v = (double) ha[i] * hx[i] + 2.0;
        want_sum += v;
        want_table[hidx[i]] += v;
    }

    CHECK(ek_hip_init(-1));
    void *a, *x, *idx, *u, *table, *sum;
    CHECK(ek_hip_malloc(n * sizeof(float), &a));
    CHECK(ek_hip_malloc(n * sizeof(float), &x));
    CHECK(ek_hip_malloc(n * sizeof(uint32_t), &idx));
    CHECK(ek_hip_malloc(n * sizeof(float), &u));
    CHECK(ek_hip_malloc(k * sizeof(float), &table));
    CHECK(ek_hip_malloc(sizeof(float), &sum));
    CHECK(ek_hip_memcpy_to_device(a, ha, n * sizeof(float)));
    CHECK(ek_hip_memcpy_to_device(x, hx, n * sizeof(float)));
    CHECK(ek_hip_memcpy_to_device(idx, hidx, n * sizeof(uint32_t)));
    CHECK(ek_hip_memset(table, 0, k * sizeof(float)));

    /* u = fmadd(a, x, 2)  -- the third operand is an immediate: ptr = NULL, bits of 2.0f in imm */
    float two = 2.0f;
    ek_operand oa = { a, 0, n }, ox = { x, 0, n }, ob = { NULL, 0, 1 }, ou = { u, 0, n }, oi = { idx, 0, n },
               om = { NULL, 1, 1 } /* mask = true */;
    memcpy(&ob.imm, &two, sizeof(float));
    CHECK(ek_hip_ternary(EK_FMADD, EK_F32, u, &oa, &ox, &ob, n));
    CHECK(ek_hip_reduce(EK_HSUM, EK_F32, sum, u, n));
    CHECK(ek_hip_scatter_add(EK_F32, EK_U32, table, k, &ou, &oi, &om, n, 0));

    float got_sum;
    static float got_table[4096];
    CHECK(ek_hip_memcpy_to_host(&got_sum, sum, sizeof(float)));
    CHECK(ek_hip_memcpy_to_host(got_table, table, k * sizeof(float)));

    int bad = ((double) got_sum != want_sum);
    for (size_t j = 0; j < k; ++j) bad += ((double) got_table[j] != want_table[j]);
    printf("hsum = %.1f (expected %.1f), scatter_add bins wrong: %d, kernel launches: %llu\n", got_sum, want_sum, bad - ((double) got_sum != want_sum),
           (unsigned long long) ek_hip_launch_count());
    CHECK(ek_hip_free(a)); CHECK(ek_hip_free(x)); CHECK(ek_hip_free(idx)); CHECK(ek_hip_free(u)); CHECK(ek_hip_free(table)); CHECK(ek_hip_free(sum));
    free(ha); free(hx); free(hidx);
    return bad ? 2 : 0;
}
