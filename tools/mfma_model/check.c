// check.c -- oracle/mfma_f16_model.h against hardware results: per-family mismatch counts, first few failing cases.
//   gcc -O2 -I oracle tools/mfma_model/check.c -o /tmp/mfma/check && /tmp/mfma/check cases.bin out.bin index.json
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mfma_f16_model.h"
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb"); int64_t n; if (!f || fread(&n, 8, 1, f) != 1) return 2;
    uint16_t* A = malloc(n * 32); uint16_t* B = malloc(n * 32); uint32_t* C = malloc(n * 4); uint32_t* D = malloc(n * 4);
    if (fread(A, 32, n, f) != (size_t)n || fread(B, 32, n, f) != (size_t)n || fread(C, 4, n, f) != (size_t)n) return 2;
    fclose(f);
    f = fopen(argv[2], "rb"); if (!f || fread(D, 4, n, f) != (size_t)n) return 2; fclose(f);
    f = fopen(argv[3], "r"); if (!f) return 2;
    char buf[65536]; size_t len = fread(buf, 1, sizeof buf - 1, f); buf[len] = 0; fclose(f);
    int show = argc > 4 ? atoi(argv[4]) : 0;
    long long bad_all = 0;
    for (char* p = buf; (p = strchr(p, '"')) != NULL;) {
        char name[64]; int i = 0; p++;
        while (*p != '"' && i < 63) name[i++] = *p++;
        name[i] = 0; p++;
        if (!strcmp(name, "start") || !strcmp(name, "n") || !strcmp(name, "per") || !strcmp(name, "d0")) continue;
        char* q = strstr(p, "\"start\":"); long long st = atoll(q + 8);
        q = strstr(p, "\"n\":"); long long cnt = atoll(q + 4);
        long long bad = 0;
        for (long long t = st; t < st + cnt; t++) {
            const uint32_t m = mfma_f16_dot16(C[t], A + t * 16, B + t * 16);
            if (m != D[t]) {
                if (bad < show) {
                    printf("  %s case %lld: hw %08x model %08x c %08x\n    a:", name, t - st, D[t], m, C[t]);
                    for (int k = 0; k < 16; k++) printf(" %04x", A[t * 16 + k]);
                    printf("\n    b:");
                    for (int k = 0; k < 16; k++) printf(" %04x", B[t * 16 + k]);
                    printf("\n");
                }
                bad++;
            }
        }
        printf("%-16s %8lld cases, %7lld differ\n", name, cnt, bad);
        bad_all += bad;
    }
    printf("total differ: %lld of %lld\n", bad_all, (long long)n);
    return 0;
}
