// sat_semantics.hip — what the VOP3 / VOP3P `clamp` bit does on gfx950 for the integer instructions a left-justified
// SAT::TCPL step could use: checked against the arithmetic definition on random and edge inputs.
//   v_add_i32 / v_sub_i32 ... clamp        : saturate(a +/- b) to int32 ?
//   v_mad_i32_i24 ... clamp                : saturate(sext24(a) * sext24(b) + c) to int32, the product at full width ?
//   v_pk_add_i16 / v_pk_sub_i16 ... clamp  : per half, saturate to int16 ?
//   v_pk_mad_i16 ... clamp                 : per half, saturate(a * b + c) to int16, the product at full width ?
//   v_add_u32 / v_mad_u32_u24 / v_pk_add_u16 / v_pk_mad_u16 ... clamp : the unsigned counterparts (saturate to 2^32 - 1 / 2^16 - 1) ?
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/sat_semantics.hip -o /tmp/sat_semantics ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k(const int* a, const int* b, const int* c, int* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = a[i], y = b[i], z = c[i];
    int r0, r1, r2, r3, r4, r5;
    asm volatile("v_add_i32 %0, %1, %2 clamp" : "=v"(r0) : "v"(x), "v"(y));
    asm volatile("v_sub_i32 %0, %1, %2 clamp" : "=v"(r1) : "v"(x), "v"(y));
    asm volatile("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(r2) : "v"(x), "v"(y), "v"(z));
    asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r3) : "v"(x), "v"(y));
    asm volatile("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r4) : "v"(x), "v"(y));
    asm volatile("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r5) : "v"(x), "v"(y), "v"(z));
    int u0, u1, u2, u3;
    asm volatile("v_add_u32 %0, %1, %2 clamp" : "=v"(u0) : "v"(x), "v"(y));
    asm volatile("v_mad_u32_u24 %0, %1, %2, %3 clamp" : "=v"(u1) : "v"(x), "v"(y), "v"(z));
    asm volatile("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(u2) : "v"(x), "v"(y));
    asm volatile("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(u3) : "v"(x), "v"(y), "v"(z));
    out[i * 10 + 0] = r0; out[i * 10 + 1] = r1; out[i * 10 + 2] = r2; out[i * 10 + 3] = r3; out[i * 10 + 4] = r4; out[i * 10 + 5] = r5;
    out[i * 10 + 6] = u0; out[i * 10 + 7] = u1; out[i * 10 + 8] = u2; out[i * 10 + 9] = u3;
}

static int64_t sat(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : v > hi ? hi : v; }
static int32_t sx24(int32_t v) { return (int32_t)((uint32_t)v << 8) >> 8; }
static int16_t lo16(int32_t v) { return (int16_t)(v & 0xffff); }
static int16_t hi16(int32_t v) { return (int16_t)((uint32_t)v >> 16); }
static int32_t pk(int64_t l, int64_t h) { return (int32_t)(((uint32_t)(uint16_t)(int16_t)h << 16) | (uint16_t)(int16_t)l); }

int main()
{
    const int n = 1 << 20;
    std::vector<int> a(n), b(n), c(n), out((size_t)n * 10);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const int edges[] = {0, 1, -1, 0x7fffffff, (int)0x80000000, 0x7fff7fff, (int)0x80008000, 0x007fffff, (int)0xff800000, 0x7fff8000, 0x00ff00ff, 255, -256, 32767, -32768};
    for (int i = 0; i < n; ++i) {
        const uint32_t m = rnd() % 8;
        a[i] = rnd(); b[i] = rnd(); c[i] = rnd();
        if (m == 0) { a[i] = edges[rnd() % 15]; b[i] = edges[rnd() % 15]; c[i] = edges[rnd() % 15]; }
        if (m == 1) { a[i] >>= 8; b[i] >>= 8; }            // 24-bit operands, products far beyond 32 bits
        if (m == 2) { a[i] >>= 16; b[i] >>= 14; }          // 16-bit x 18-bit
        if (m == 3) { a[i] = pk((int16_t)rnd() >> 4, (int16_t)rnd() >> 3); b[i] = pk((int16_t)rnd() >> 9, (int16_t)rnd() >> 8); c[i] = pk((int16_t)rnd() >> 1, (int16_t)rnd() >> 2); }
        if (m == 4) { a[i] >>= 1; b[i] >>= 1; }
    }
    int *da, *db, *dc, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, (size_t)n * 40);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    if (hipMemcpy(out.data(), dout, (size_t)n * 40, hipMemcpyDeviceToHost) != hipSuccess) { printf("{\"error\": \"hip\"}\n"); return 1; }
    long bad[6] = {0, 0, 0, 0, 0, 0}, sat_hits[6] = {0, 0, 0, 0, 0, 0}, ubad[4] = {0, 0, 0, 0};
    int ufirst[4] = {-1, -1, -1, -1};
    int first[6] = {-1, -1, -1, -1, -1, -1};
    const int64_t L32 = INT32_MIN, H32 = INT32_MAX;
    for (int i = 0; i < n; ++i) {
        const int64_t x = a[i], y = b[i], z = c[i];
        int32_t e[6];
        e[0] = (int32_t)sat(x + y, L32, H32);
        e[1] = (int32_t)sat(x - y, L32, H32);
        e[2] = (int32_t)sat((int64_t)sx24(a[i]) * sx24(b[i]) + z, L32, H32);
        e[3] = pk(sat((int64_t)lo16(a[i]) + lo16(b[i]), -32768, 32767), sat((int64_t)hi16(a[i]) + hi16(b[i]), -32768, 32767));
        e[4] = pk(sat((int64_t)lo16(a[i]) - lo16(b[i]), -32768, 32767), sat((int64_t)hi16(a[i]) - hi16(b[i]), -32768, 32767));
        e[5] = pk(sat((int64_t)lo16(a[i]) * lo16(b[i]) + lo16(c[i]), -32768, 32767), sat((int64_t)hi16(a[i]) * hi16(b[i]) + hi16(c[i]), -32768, 32767));
        const int64_t raw[6] = {x + y, x - y, (int64_t)sx24(a[i]) * sx24(b[i]) + z, (int64_t)lo16(a[i]) + lo16(b[i]), (int64_t)lo16(a[i]) - lo16(b[i]), (int64_t)lo16(a[i]) * lo16(b[i]) + lo16(c[i])};
        for (int j = 0; j < 6; ++j) {
            const bool s32 = j < 3 ? (raw[j] < L32 || raw[j] > H32) : (raw[j] < -32768 || raw[j] > 32767);
            sat_hits[j] += s32;
            if (out[(size_t)i * 10 + j] != e[j]) { if (first[j] < 0) first[j] = i; ++bad[j]; }
        }
        // unsigned: saturate to 2^32 - 1 / 2^16 - 1 per half; the u24 multiply takes the low 24 bits of its operands
        {
            const uint64_t ux = (uint32_t)a[i], uy = (uint32_t)b[i], uz = (uint32_t)c[i];
            auto usat = [](uint64_t v, uint64_t hi) { return v > hi ? hi : v; };
            auto l16 = [](uint64_t v) { return v & 0xffffu; };
            auto h16 = [](uint64_t v) { return (v >> 16) & 0xffffu; };
            const uint32_t ue[4] = {(uint32_t)usat(ux + uy, 0xffffffffull), (uint32_t)usat((ux & 0xffffff) * (uy & 0xffffff) + uz, 0xffffffffull),
                                    (uint32_t)(usat(l16(ux) + l16(uy), 0xffff) | (usat(h16(ux) + h16(uy), 0xffff) << 16)),
                                    (uint32_t)(usat(l16(ux) * l16(uy) + l16(uz), 0xffff) | (usat(h16(ux) * h16(uy) + h16(uz), 0xffff) << 16))};
            for (int j = 0; j < 4; ++j)
                if ((uint32_t)out[(size_t)i * 10 + 6 + j] != ue[j]) { if (ufirst[j] < 0) ufirst[j] = i; ++ubad[j]; }
        }
    }
    const char* names[6] = {"v_add_i32 clamp", "v_sub_i32 clamp", "v_mad_i32_i24 clamp", "v_pk_add_i16 clamp", "v_pk_sub_i16 clamp", "v_pk_mad_i16 clamp"};
    printf("{\"cases\": %d", n);
    for (int j = 0; j < 6; ++j) {
        printf(", \"%s\": {\"mismatches\": %ld, \"saturating_cases\": %ld", names[j], bad[j], sat_hits[j]);
        if (first[j] >= 0) { const int i = first[j]; printf(", \"first\": [%d, %d, %d, %d]", a[i], b[i], c[i], out[(size_t)i * 6 + j]); }
        printf("}");
    }
    const char* unames[4] = {"v_add_u32 clamp", "v_mad_u32_u24 clamp", "v_pk_add_u16 clamp", "v_pk_mad_u16 clamp"};
    for (int j = 0; j < 4; ++j) {
        printf(", \"%s\": {\"mismatches\": %ld", unames[j], ubad[j]);
        if (ufirst[j] >= 0) { const int i = ufirst[j]; printf(", \"first\": [%d, %d, %d, %d]", a[i], b[i], c[i], out[(size_t)i * 10 + 6 + j]); }
        printf("}");
    }
    printf("}\n");
    return 0;
}
