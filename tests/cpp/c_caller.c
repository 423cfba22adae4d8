/* A plain C99 caller of include/vdf.h - what a cgo / bindgen / ctypes binding sees: the header must be C (not C++), and the host-only
 * entry points must work without a GPU.  Built and run by tests/test_capi_symbols.py (gcc -std=c99 -pedantic -Werror). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vdf.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "c_caller: %s failed (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main(void)
{
    uint64_t a[VDF_HASH_WORDS], b[VDF_HASH_WORDS];
    int i;
    for (i = 0; i < VDF_HASH_WORDS; i++) { a[i] = 0; b[i] = ~(uint64_t)0; }
    CHECK(vdf_hamming_u1024(a, a) == 0 && vdf_hamming_u1024(a, b) == 1024);          /* video_hash.rs:311-317: all 16 words */
    CHECK(vdf_tolerance_int(0.35) == 350);                                           /* DEFAULT_SEARCH_TOLERANCE x TOLERANCE_SCALING_FACTOR */
    CHECK(vdf_version() != NULL && strlen(vdf_version()) > 0);

    /* no GPU: a context is refused with a status and a message, never a crash or a CPU stand-in */
    {
        vdf_ctx *ctx = NULL;
        const int rc = vdf_ctx_create(0, &ctx);
        if (rc == VDF_OK) {
            CHECK(ctx != NULL && vdf_ctx_device(ctx) == 0);
            vdf_ctx_destroy(ctx);
        } else {
            CHECK(rc == VDF_E_HIP && ctx == NULL && strlen(vdf_last_error(NULL)) > 0);
        }
    }

    /* the cache codec and the sidecar are host-only */
    {
        const char paths[] = "a/bc";  /* two paths: "a" and "/bc" */
        const uint64_t offs[3] = {0, 1, 4};
        uint64_t hashes[2 * VDF_HASH_WORDS];
        const uint32_t durs[2] = {7, 4000000000u};
        uint8_t *bytes = NULL;
        size_t len = 0;
        vdf_cache_soa soa;
        vdf_cache_metadata md;
        char text[128];
        size_t text_len = 0;
        uint32_t rank[2];
        for (i = 0; i < 2 * VDF_HASH_WORDS; i++) hashes[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        CHECK(vdf_cache_encode(2, hashes, durs, offs, paths, NULL, NULL, &bytes, &len) == VDF_OK && bytes != NULL && len > 2 * 128);
        CHECK(vdf_cache_decode(bytes, len, &soa) == VDF_OK && soa.n_ok == 2 && soa.n_err == 0);
        CHECK(memcmp(soa.hashes, hashes, sizeof hashes) == 0 && soa.durations[1] == 4000000000u && soa.path_offsets[2] == 4);
        CHECK(vdf_cache_decode(bytes, len - 1, &soa) == VDF_E_INVAL);
        vdf_buffer_free(bytes);
        CHECK(vdf_cache_decode(NULL, 0, &soa) == VDF_E_INVAL);
        CHECK(vdf_cache_metadata_new(VDF_CROPDETECT_LETTERBOX, 15.0, &md) == VDF_OK);
        CHECK(vdf_cache_metadata_format(&md, text, sizeof text, &text_len) == VDF_OK && strcmp(text, "Unix,FfmpegBackend,Letterbox,15,1") == 0);
        CHECK(vdf_path_compare("a/b", 3, "a.b", 3) < 0);                                 /* components, not bytes */
        CHECK(vdf_path_ranks(paths, offs, 2, rank, 1) == VDF_OK && rank[0] == 1 && rank[1] == 0);  /* RootDir sorts before Normal: "/bc" < "a" */
    }

    /* the greedy replay (search_algorithm.rs:131-170) on a hand-made adjacency: 0-1, 0-2, 3-4 */
    {
        const vdf_hit hits[3] = {{0, 1}, {0, 2}, {3, 4}};
        vdf_groups g;
        memset(&g, 0, sizeof g);
        CHECK(vdf_replay_self(5, hits, 3, 0, 5, NULL, &g) == VDF_OK && vdf_groups_finish_self(&g) == VDF_OK);
        CHECK(g.n_groups == 2 && g.offsets[2] == 5);
        vdf_groups_free(&g);
    }
    puts("c caller ok");
    return 0;
}
