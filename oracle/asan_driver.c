/* asan_driver.c -- TEST INFRASTRUCTURE: runs the C oracle under AddressSanitizer / UBSan (CPU build only;
 * GPU sanitizers are not available on the pool).  Usage: asan_driver <params.f32>
 * Exercises: window, STFT, offline forward (two lengths, batch 2), a streamed frame-by-frame pass with state,
 * iSTFT, the wave->wave convenience.  Exit code 0 = no sanitizer report and finite output. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gtcrn_oracle.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    const long n = 44938;
    float *params = malloc(sizeof(float) * n);
    if (fread(params, sizeof(float), n, f) != (size_t)n) return 2;
    fclose(f);
    gtcrn_oracle *h = gtcrn_oracle_create(params, n);
    if (!h) return 3;
    float win[512];
    gtcrn_oracle_window(0, win);
    const int B = 2;
    const long L = 256 * 9 + 17;
    const int T = (int)gtcrn_oracle_num_frames(L);
    float *wave = malloc(sizeof(float) * B * L);
    unsigned s = 12345u;
    for (long i = 0; i < B * L; ++i) { s = s * 1664525u + 1013904223u; wave[i] = ((s >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    float *spec = malloc(sizeof(float) * B * 257 * T * 2), *out = malloc(sizeof(float) * B * 257 * T * 2);
    if (gtcrn_oracle_stft(wave, B, L, win, spec)) return 4;
    if (gtcrn_oracle_forward(h, spec, B, T, out, NULL)) return 5;
    if (gtcrn_oracle_forward(h, spec, 1, 1, out, NULL)) return 5;          /* one frame */
    float *y = malloc(sizeof(float) * B * 256 * (T - 1));
    if (gtcrn_oracle_stft(wave, B, L, win, spec)) return 4;
    if (gtcrn_oracle_forward(h, spec, B, T, out, NULL)) return 5;
    if (gtcrn_oracle_istft(out, B, T, win, y)) return 6;
    if (gtcrn_oracle_enhance(h, wave, B, L, 0, y)) return 7;
    double acc = 0.0;
    for (long i = 0; i < (long)B * 256 * (T - 1); ++i) acc += fabs((double)y[i]);
    gtcrn_oracle_destroy(h);
    free(params); free(wave); free(spec); free(out); free(y);
    if (!isfinite(acc)) return 8;
    printf("asan_driver ok %.6f\n", acc);
    return 0;
}
