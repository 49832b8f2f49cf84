/*
 * o_tiled.c — oracle restatement of TiledImage's sparsity rule.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows
 *   src/canvas/tiled_image.rs:50-104  from_rgba_image (chunk kept iff any alpha != 0)
 *   src/canvas/tiled_image.rs:271-293 to_rgba_image   (missing chunk -> zeros)
 */
#include "o_common.h"

void pfxo_chunk_populated(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* populated)
{
    uint32_t cxn = (w + PFXO_CHUNK - 1) / PFXO_CHUNK, cyn = (h + PFXO_CHUNK - 1) / PFXO_CHUNK;
    memset(populated, 0, (size_t)cxn * cyn);
#pragma omp parallel for schedule(static)
    for (long cy = 0; cy < (long)cyn; ++cy) {
        uint32_t y1 = (uint32_t)(cy + 1) * PFXO_CHUNK;
        if (y1 > h) y1 = h;
        for (uint32_t y = (uint32_t)cy * PFXO_CHUNK; y < y1; ++y) {
            const uint8_t* row = rgba + (size_t)y * w * 4;
            for (uint32_t x = 0; x < w; ++x)
                if (row[(size_t)x * 4 + 3] != 0) populated[(size_t)cy * cxn + x / PFXO_CHUNK] = 1;
        }
    }
}

void pfxo_tiled_roundtrip(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst)
{
    uint32_t cxn = (w + PFXO_CHUNK - 1) / PFXO_CHUNK, cyn = (h + PFXO_CHUNK - 1) / PFXO_CHUNK;
    uint8_t* pop = (uint8_t*)malloc((size_t)cxn * cyn);
    pfxo_chunk_populated(src, w, h, pop);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            size_t i = ((size_t)y * w + x) * 4;
            if (pop[o_chunk_index(x, y, w)]) memcpy(dst + i, src + i, 4);
            else memset(dst + i, 0, 4);
        }
    free(pop);
}

int pfxo_version(void) { return 1; }
