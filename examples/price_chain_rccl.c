/*
 * price_chain_rccl.c -- a plain-C multi-GPU host of libsvmc.so: one process per GPU, paths sharded by global path id,
 * the two reductions of a chain as RCCL all-reduces issued by the fused chain driver itself (include/svmc.h,
 * "multi-GPU below the host language").  No Python, no MPI: rank 0 writes the RCCL unique id to a file, the other
 * ranks pick it up.
 *
 *   gcc -O2 -Iinclude examples/price_chain_rccl.c -o price_chain_rccl -Lstochvolmodels_amd -lsvmc \
 *       -Wl,-rpath,$PWD/stochvolmodels_amd -lm
 *   for r in 0 1 2 3 4 5 6 7; do ./price_chain_rccl 8 $r /tmp/svmc_id 16777216 & done; wait
 *
 * Every rank prints the job's prices (identical on all ranks, and identical to the single-GPU result up to the
 * order of the final sums).  With world = 1 the numbers are those of examples/price_chain.c / the Python host bit for
 * bit: tests/test_gpu_parity.py::test_c_host_rccl_example.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
    if (argc < 4) {
        fprintf(stderr, "usage: %s world rank id_file [n_path_total] [seed]\n", argv[0]);
        return 2;
    }
    const int world = atoi(argv[1]), rank = atoi(argv[2]);
    const char *id_file = argv[3];
    const uint64_t n_total = (argc > 4) ? strtoull(argv[4], NULL, 10) : 65536;
    const uint64_t seed = (argc > 5) ? strtoull(argv[5], NULL, 10) : 20240601ull;

    /* fail fast: a rank that dies before ncclCommInitRank leaves the others waiting inside it for ever -- SIGALRM's default
     * action ends this process (non-zero status) when the whole program has not finished within its deadline */
    alarm(getenv("SVMC_EXAMPLE_DEADLINE") ? (unsigned)atoi(getenv("SVMC_EXAMPLE_DEADLINE")) : 300u);

    int n_dev = 0;
    CHECK(svmc_device_count(&n_dev));
    CHECK(svmc_set_device(rank % n_dev));

    /* the RCCL unique id: rank 0 makes it and publishes it through the file system (write + rename = atomic).  The file
     * carries the run's token (here: the seed all ranks were started with) in front of the id, and a reader accepts only a
     * file with ITS token: an id file left behind by a crashed earlier run (rank 0 removes the file only on success) is
     * neither mistaken for this run's -- the reader keeps waiting for the right one -- nor left in place by rank 0 */
    unsigned char id[SVMC_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        unlink(id_file);
        CHECK(svmc_rccl_unique_id(id, sizeof id));
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *f = fopen(tmp, "wb");
        if (f == NULL || fwrite(&seed, sizeof seed, 1, f) != 1 || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 ||
            rename(tmp, id_file) != 0) {
            perror("publishing the RCCL id");
            return 1;
        }
    } else {
        int have = 0;
        for (int tries = 0; tries < 600 && !have; ++tries) {
            FILE *f = fopen(id_file, "rb");
            uint64_t token = 0;
            if (f != NULL) {
                have = fread(&token, sizeof token, 1, f) == 1 && token == seed && fread(id, 1, sizeof id, f) == sizeof id;
                fclose(f);
            }
            if (!have) usleep(100000);
        }
        if (!have) {
            fprintf(stderr, "rank %d: no RCCL id of this run (token %llu) at %s\n", rank, (unsigned long long)seed, id_file);
            return 1;
        }
    }
    svmc_comm_t comm;
    CHECK(svmc_rccl_comm_create(&comm, id, sizeof id, world, rank));

    /* this rank's paths: [lo, hi) of the job's global path ids */
    const uint64_t lo = n_total * (uint64_t)rank / (uint64_t)world, hi = n_total * (uint64_t)(rank + 1) / (uint64_t)world;
    svmc_session_t session;
    CHECK(svmc_session_create(&session, (size_t)(hi - lo), 2, 6));
    CHECK(svmc_session_set_comm(session, comm, rank, world, n_total, lo));

    /* the chain of examples/price_chain.c */
    const double ttms[2] = {0.1, 0.25}, forwards[2] = {1.0, 1.01}, discfactors[2] = {0.99, 0.98};
    const double strikes[6] = {0.8, 1.0, 1.2, 0.8 * 1.01, 1.0 * 1.01, 1.2 * 1.01};
    const int8_t types[6] = {SVMC_PUT, SVMC_CALL, SVMC_CALL, SVMC_INV_PUT, SVMC_INV_CALL, SVMC_CALL};
    const size_t offsets[3] = {0, 3, 6};
    double prices[6], stderrs[6], hprices[6], hstderrs[6];
    CHECK(svmc_logsv_chain_price(session, ttms, forwards, discfactors, NULL, 2, strikes, types, offsets, 0.8376, 1.0413,
                                 3.1844, 3.058, 0.1514, 1.8458, 1, 120, SVMC_LOG_RETURN, seed, 0, prices, stderrs));
    CHECK(svmc_heston_chain_price(session, ttms, forwards, discfactors, 2, strikes, types, offsets, 0.04, 0.04, 4.0, -0.5,
                                  0.4, SVMC_HESTON_QE, 360, SVMC_LOG_RETURN, seed, 0, hprices, hstderrs));
    printf("{\"rank\": %d, \"world\": %d, \"n_path_total\": %llu, \"n_path_local\": %llu, \"rccl\": \"%s\", ", rank, world,
           (unsigned long long)n_total, (unsigned long long)(hi - lo), svmc_rccl_origin());
    print_array("logsv_prices", prices, 6, 0);
    print_array("logsv_stderrs", stderrs, 6, 0);
    print_array("heston_qe_prices", hprices, 6, 0);
    print_array("heston_qe_stderrs", hstderrs, 6, 1);
    printf("}\n");
    fflush(stdout);

    CHECK(svmc_session_destroy(session));
    CHECK(svmc_rccl_comm_destroy(comm));
    if (rank == 0) unlink(id_file);
    return 0;
}
