/*
 * price_chain.c -- a plain-C host of libsvmc.so: prices a two-expiry LogSV chain and a one-expiry Heston chain by
 * Monte Carlo on the GPU through the fused chain drivers of include/svmc.h, re-prices one expiry on fixed randoms
 * resident in HBM (the calibration inner loop), prices the LogSV chain once more on state arrays it owns itself
 * (svmc_session_create_on) and prints the results as JSON.
 *
 *   gcc -O2 -Iinclude examples/price_chain.c -o price_chain -Lstochvolmodels_amd -lsvmc \
 *       -Wl,-rpath,$PWD/stochvolmodels_amd -lm
 *   ./price_chain [n_path] [seed]
 *
 * The same chain priced through the Python host (stochvolmodels_amd.logsv_mc_chain_pricer with the same seed) gives
 * the same numbers: tests/test_gpu_parity.py::test_c_host_example.
 */
#include <stdio.h>
#include <stdlib.h>

#include "svmc.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != SVMC_OK) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, svmc_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static void print_array(const char *name, const double *a, size_t n, int last)
{
    printf("\"%s\": [", name);
    for (size_t i = 0; i < n; ++i) printf("%s%.17g", i ? ", " : "", a[i]);
    printf("]%s", last ? "" : ", ");
}

int main(int argc, char **argv)
{
    const size_t n_path = (argc > 1) ? (size_t)strtoull(argv[1], NULL, 10) : 65536;
    const uint64_t seed = (argc > 2) ? strtoull(argv[2], NULL, 10) : 20240601ull;

    int n_dev = 0;
    CHECK(svmc_device_count(&n_dev));
    CHECK(svmc_set_device(0));

    /* chain: ttms 0.1 and 0.25; strikes 0.8, 1.0, 1.2 x forward; P, C, C then IP, IC, C */
    const double ttms[2] = {0.1, 0.25}, forwards[2] = {1.0, 1.01}, discfactors[2] = {0.99, 0.98};
    const double strikes[6] = {0.8, 1.0, 1.2, 0.8 * 1.01, 1.0 * 1.01, 1.2 * 1.01};
    const int8_t types[6] = {SVMC_PUT, SVMC_CALL, SVMC_CALL, SVMC_INV_PUT, SVMC_INV_CALL, SVMC_CALL};
    const size_t offsets[3] = {0, 3, 6};
    double prices[6], stderrs[6];

    svmc_session_t session;
    CHECK(svmc_session_create(&session, n_path, 2, 6));

    /* LOGSV_BTC_PARAMS of the reference (pricers/logsv_pricer.py:102) */
    CHECK(svmc_logsv_chain_price(session, ttms, forwards, discfactors, NULL, 2, strikes, types, offsets,
                                 0.8376, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, /*spot measure*/ 1,
                                 /*steps per year*/ 120, SVMC_LOG_RETURN, seed, 0, prices, stderrs));
    printf("{\"svmc_version\": %d, \"n_path\": %zu, ", svmc_version(), n_path);
    print_array("logsv_prices", prices, 6, 0);
    print_array("logsv_stderrs", stderrs, 6, 0);

    /* Heston, the reference's default parameters (pricers/heston_pricer.py:36-40), Euler scheme, first expiry only */
    CHECK(svmc_heston_chain_price(session, ttms, forwards, discfactors, 1, strikes, types, offsets, 0.04, 0.04, 4.0, -0.5,
                                  0.4, SVMC_HESTON_EULER_FLOOR, 360, SVMC_LOG_RETURN, seed, 0, prices, stderrs));
    print_array("heston_prices", prices, 3, 0);
    print_array("heston_stderrs", stderrs, 3, 0);

    /* The inner loop of an MC calibration: fixed N(0,1) randoms resident in HBM (here drawn on the device; a
     * calibration uploads its own with svmc_memcpy2d_h2d), the first expiry re-priced on them for two parameter
     * sets with svmc_logsv_chain_price_fixed, and the call's price inverted to a Black vol on the host. */
    const int nb_steps[1] = {12};
    const double dts[1] = {0.1 / 12};
    double *w0 = NULL, *w1 = NULL;
    CHECK(svmc_malloc((void **)&w0, sizeof(double) * n_path * 12));
    CHECK(svmc_malloc((void **)&w1, sizeof(double) * n_path * 12));
    CHECK(svmc_fill_normals(w0, w1, n_path, n_path, 12, seed, 1, 0, 0, NULL));
    CHECK(svmc_stream_synchronize(NULL));
    const double *w0s[1] = {w0}, *w1s[1] = {w1};
    double fixed_prices[6], ivol[1];
    for (int it = 0; it < 2; ++it) {
        const double volvol = it ? 1.2 : 1.8458;
        CHECK(svmc_logsv_chain_price_fixed(session, ttms, forwards, discfactors, NULL, 1, strikes, types, offsets, 0.8376,
                                           1.0413, 3.1844, 3.058, 0.1514, volvol, 1, SVMC_LOG_RETURN, w0s, w1s, nb_steps,
                                           dts, n_path, fixed_prices + 3 * it, stderrs));
    }
    print_array("fixed_prices", fixed_prices, 6, 0);
    CHECK(svmc_black_implied_vols(fixed_prices + 1, strikes + 1, types + 1, 1, forwards[0], ttms[0], discfactors[0], 1e-6,
                                  10.0, ivol));
    print_array("atm_call_ivol", ivol, 1, 0);
    CHECK(svmc_free(w0));
    CHECK(svmc_free(w1));

    /* A host that keeps the state arrays itself (svmc_session_create_on: what the Python host does on one GPU): the session
     * allocates only its snapshots and scratch, the fused driver leaves the terminal (x, sigma, qvar) of every path in the
     * caller's arrays, and destroying the session frees neither them nor the stream.  Same kernels, same bits. */
    double *x = NULL, *sigma = NULL, *qvar = NULL, own_prices[6], own_stderrs[6], head[3][4];
    svmc_session_t on_mine;
    CHECK(svmc_malloc((void **)&x, sizeof(double) * n_path));
    CHECK(svmc_malloc((void **)&sigma, sizeof(double) * n_path));
    CHECK(svmc_malloc((void **)&qvar, sizeof(double) * n_path));
    CHECK(svmc_session_create_on(&on_mine, n_path, 2, 6, x, sigma, qvar, /*path_offset*/ 0, /*stream*/ NULL));
    CHECK(svmc_logsv_chain_price(on_mine, ttms, forwards, discfactors, NULL, 2, strikes, types, offsets, 0.8376, 1.0413, 3.1844,
                                 3.058, 0.1514, 1.8458, 1, 120, SVMC_LOG_RETURN, seed, 0, own_prices, own_stderrs));
    CHECK(svmc_session_destroy(on_mine));
    CHECK(svmc_memcpy_d2h(head[0], x, sizeof(double) * 4, NULL));          /* still ours, and holding the terminal state */
    CHECK(svmc_memcpy_d2h(head[1], sigma, sizeof(double) * 4, NULL));
    CHECK(svmc_memcpy_d2h(head[2], qvar, sizeof(double) * 4, NULL));
    CHECK(svmc_stream_synchronize(NULL));
    print_array("logsv_prices_on_caller_state", own_prices, 6, 0);
    print_array("terminal_x_head", head[0], 4, 0);
    print_array("terminal_sigma_head", head[1], 4, 0);
    print_array("terminal_qvar_head", head[2], 4, 1);
    printf("}\n");
    CHECK(svmc_free(x));
    CHECK(svmc_free(sigma));
    CHECK(svmc_free(qvar));

    CHECK(svmc_session_destroy(session));
    return 0;
}
