/*
 * price_chain_multi.c -- a plain-C host that uses SEVERAL GPUs from ONE process: the chain of examples/price_chain.c on a
 * multi-session (include/svmc.h, "single-process multi-device sessions").  No launcher, no rendezvous, no id file: the
 * library runs one host thread and one session per shard and issues the two all-reduces of a chain itself (RCCL where it
 * resolves and the shards sit on distinct devices, else a sum through pinned host memory).
 *
 *   gcc -O2 -Iinclude examples/price_chain_multi.c -o price_chain_multi -Lstochvolmodels_amd -lsvmc \
 *       -Wl,-rpath,$PWD/stochvolmodels_amd -lm
 *   ./price_chain_multi 8 16777216            # 8 shards on devices 0..7 (shard r on device r mod the visible devices)
 *   ./price_chain_multi 3 65536 20240601 host # the host transport, e.g. three shards on one device
 *
 * With one shard the numbers are those of examples/price_chain.c bit for bit; with R shards they differ from it by the
 * order of the final additions only (tests/test_gpu_multi.py::test_c_host_multi_example).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    const int n_shards = (argc > 1) ? atoi(argv[1]) : 2;
    const uint64_t n_total = (argc > 2) ? strtoull(argv[2], NULL, 10) : 65536;
    const uint64_t seed = (argc > 3) ? strtoull(argv[3], NULL, 10) : 20240601ull;
    int mode = SVMC_MULTI_REDUCE_AUTO;
    if (argc > 4 && strcmp(argv[4], "host") == 0) mode = SVMC_MULTI_REDUCE_HOST;
    if (argc > 4 && strcmp(argv[4], "rccl") == 0) mode = SVMC_MULTI_REDUCE_RCCL;

    svmc_multi_t multi;
    CHECK(svmc_multi_create(&multi, n_shards, NULL, n_total, 2, 6, mode));

    const double ttms[2] = {0.1, 0.25}, forwards[2] = {1.0, 1.01}, discfactors[2] = {0.99, 0.98};
    const double strikes[6] = {0.8, 1.0, 1.2, 0.8 * 1.01, 1.0 * 1.01, 1.2 * 1.01};
    const int8_t types[6] = {SVMC_PUT, SVMC_CALL, SVMC_CALL, SVMC_INV_PUT, SVMC_INV_CALL, SVMC_CALL};
    const size_t offsets[3] = {0, 3, 6};
    double prices[6], stderrs[6], hprices[6], hstderrs[6];
    /* LOGSV_BTC_PARAMS of the reference (pricers/logsv_pricer.py:102); Heston QE on the reference's default parameters */
    CHECK(svmc_multi_logsv_chain_price(multi, ttms, forwards, discfactors, NULL, 2, strikes, types, offsets, 0.8376, 1.0413,
                                       3.1844, 3.058, 0.1514, 1.8458, 1, 120, SVMC_LOG_RETURN, seed, 0, prices, stderrs));
    int agree_logsv = 0, used = 0, seen = 0;
    CHECK(svmc_multi_info(multi, NULL, &used, &seen, &agree_logsv));
    CHECK(svmc_multi_heston_chain_price(multi, ttms, forwards, discfactors, 2, strikes, types, offsets, 0.04, 0.04, 4.0, -0.5,
                                        0.4, SVMC_HESTON_QE, 360, SVMC_LOG_RETURN, seed, 0, hprices, hstderrs));
    int agree_heston = 0;
    CHECK(svmc_multi_info(multi, NULL, NULL, NULL, &agree_heston));

    printf("{\"n_shards\": %d, \"n_path_total\": %llu, \"reduce\": \"%s\", \"rccl_ranks_seen\": %d, \"shards_agree\": %s, ",
           n_shards, (unsigned long long)n_total, used == SVMC_MULTI_REDUCE_RCCL ? "rccl" : "host", seen,
           (agree_logsv && agree_heston) ? "true" : "false");
    printf("\"shards\": [");
    for (int r = 0; r < n_shards; ++r) {
        int dev = 0;
        uint64_t off = 0, cnt = 0;
        double ms = 0.0;
        CHECK(svmc_multi_shard_info(multi, r, &dev, &off, &cnt, &ms));
        printf("%s{\"device\": %d, \"path_offset\": %llu, \"n_path\": %llu, \"last_call_ms\": %.3f}", r ? ", " : "", dev,
               (unsigned long long)off, (unsigned long long)cnt, ms);
    }
    printf("], ");
    print_array("logsv_prices", prices, 6, 0);
    print_array("logsv_stderrs", stderrs, 6, 0);
    print_array("heston_qe_prices", hprices, 6, 0);
    print_array("heston_qe_stderrs", hstderrs, 6, 1);
    printf("}\n");
    CHECK(svmc_multi_destroy(multi));
    return (agree_logsv && agree_heston) ? 0 : 3;
}
