// dev micro-benchmark: ways to copy a 1.5 MB frame into a (pinned) staging buffer
#define _GNU_SOURCE
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
void copy_memcpy(void* d, const void* s, size_t n) { memcpy(d, s, n); }
void copy_nt(void* d, const void* s, size_t n) {
    char* dd = (char*)d; const char* ss = (const char*)s;
    while (((uintptr_t)dd & 31) && n) { *dd++ = *ss++; --n; }
    size_t v = n / 32;
    for (size_t i = 0; i < v; ++i) _mm256_stream_si256((__m256i*)dd + i, _mm256_loadu_si256((const __m256i*)ss + i));
    _mm_sfence();
    memcpy(dd + v * 32, ss + v * 32, n - v * 32);
}
typedef struct { void* d; const void* s; size_t n; volatile int go; volatile int done; int nt; } job_t;
static job_t jobs[8]; static pthread_t th[8]; static int nth = 0; static volatile int quit = 0;
static void* worker(void* a) { job_t* j = (job_t*)a; for (;;) { while (!j->go) { if (quit) return 0; _mm_pause(); } j->go = 0; if (j->nt) copy_nt(j->d, j->s, j->n); else memcpy(j->d, j->s, j->n); __sync_synchronize(); j->done = 1; } }
void pool_start(int n) { nth = n; for (int i = 0; i < n; ++i) { jobs[i].go = 0; jobs[i].done = 0; pthread_create(&th[i], 0, worker, &jobs[i]); } }
void copy_pool(void* d, const void* s, size_t n, int nt) {
    int parts = nth + 1; size_t chunk = (n / parts) & ~(size_t)63;
    for (int i = 0; i < nth; ++i) { jobs[i].d = (char*)d + (i + 1) * chunk; jobs[i].s = (const char*)s + (i + 1) * chunk; jobs[i].n = (i == nth - 1) ? n - (i + 1) * chunk : chunk; jobs[i].nt = nt; jobs[i].done = 0; __sync_synchronize(); jobs[i].go = 1; }
    if (nt) copy_nt(d, s, chunk); else memcpy(d, s, chunk);
    for (int i = 0; i < nth; ++i) while (!jobs[i].done) _mm_pause();
}
int main(int argc, char** argv) {
    size_t n = 131072 * 12; int reps = 200;
    char* src[8]; for (int i = 0; i < 8; ++i) { src[i] = (char*)malloc(n); memset(src[i], i + 1, n); }
    char* dst = (char*)aligned_alloc(4096, n); memset(dst, 0, n);
    char* junk = (char*)malloc(64 << 20);
    for (int mode = 0; mode < 6; ++mode) {
        if (mode == 2) pool_start(1); 
        if (mode == 4) { quit = 1; for (int i = 0; i < nth; ++i) pthread_join(th[i], 0); quit = 0; pool_start(3); }
        double best = 1e9, sum = 0;
        for (int r = 0; r < reps; ++r) {
            memset(junk, r, 64 << 20);  // evict
            double t0 = now();
            if (mode == 0) copy_memcpy(dst, src[r & 7], n);
            else if (mode == 1) copy_nt(dst, src[r & 7], n);
            else if (mode == 2) copy_pool(dst, src[r & 7], n, 0);
            else if (mode == 3) copy_pool(dst, src[r & 7], n, 1);
            else if (mode == 4) copy_pool(dst, src[r & 7], n, 0);
            else copy_pool(dst, src[r & 7], n, 1);
            double t = now() - t0; if (t < best) best = t; sum += t;
        }
        const char* names[] = {"memcpy", "nt stores", "2 threads memcpy", "2 threads nt", "4 threads memcpy", "4 threads nt"};
        printf("%-18s mean %.1f us  best %.1f us  (%.1f GB/s)\n", names[mode], sum / reps, best, n / (sum / reps) / 1e3);
    }
    quit = 1; return 0;
}
