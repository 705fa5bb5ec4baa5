#include <immintrin.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
typedef __m512 (*f1_t)(__m512);
typedef __m512 (*f2_t)(__m512, __m512);
typedef __m256 (*g1_t)(__m256);
typedef __m256 (*g2_t)(__m256, __m256);
static void* H;
static void* sym(const char* n) { if (!H) H = dlopen("/usr/local/lib/python3.10/dist-packages/torch/lib/libtorch_cpu.so", RTLD_NOW | RTLD_NOLOAD); if (!H) H = dlopen("/usr/local/lib/python3.10/dist-packages/torch/lib/libtorch_cpu.so", RTLD_NOW); return dlsym(H, n); }
int sleef1_16(const char* name, const float* x, float* y, int n) {
    f1_t f = (f1_t)sym(name); if (!f) return 1;
    for (int i = 0; i < n; i += 16) { float t[16] = {0}, o[16]; int m = n - i < 16 ? n - i : 16; memcpy(t, x + i, m * 4); _mm512_storeu_ps(o, f(_mm512_loadu_ps(t))); memcpy(y + i, o, m * 4); }
    return 0;
}
int sleef2_16(const char* name, const float* x, const float* p, float* y, int n) {
    f2_t f = (f2_t)sym(name); if (!f) return 1;
    for (int i = 0; i < n; i += 16) { float t[16] = {0}, q[16] = {0}, o[16]; int m = n - i < 16 ? n - i : 16; memcpy(t, x + i, m * 4); memcpy(q, p + i, m * 4); _mm512_storeu_ps(o, f(_mm512_loadu_ps(t), _mm512_loadu_ps(q))); memcpy(y + i, o, m * 4); }
    return 0;
}
int sleef1_8(const char* name, const float* x, float* y, int n) {
    g1_t f = (g1_t)sym(name); if (!f) return 1;
    for (int i = 0; i < n; i += 8) { float t[8] = {0}, o[8]; int m = n - i < 8 ? n - i : 8; memcpy(t, x + i, m * 4); _mm256_storeu_ps(o, f(_mm256_loadu_ps(t))); memcpy(y + i, o, m * 4); }
    return 0;
}
int sleef2_8(const char* name, const float* x, const float* p, float* y, int n) {
    g2_t f = (g2_t)sym(name); if (!f) return 1;
    for (int i = 0; i < n; i += 8) { float t[8] = {0}, q[8] = {0}, o[8]; int m = n - i < 8 ? n - i : 8; memcpy(t, x + i, m * 4); memcpy(q, p + i, m * 4); _mm256_storeu_ps(o, f(_mm256_loadu_ps(t), _mm256_loadu_ps(q))); memcpy(y + i, o, m * 4); }
    return 0;
}
