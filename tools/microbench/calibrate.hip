// calibrate.hip — what do the PMC counters behind bench.py's roofline count, and what are the peaks they are compared with?
//   Every dispatch here has a KNOWN number of wave-instructions / lane-loads; run once plain for the
// times and once per `rocprofv3 --pmc` pass for the counters, then tools/calibration_summary.py joins the two by dispatch order.
//
//   VALU: INDEPENDENT instruction streams (16 accumulators per lane, no dependence between neighbouring instructions), 6 waves per
//         SIMD resident (256 CUs x 6 workgroups of 256), one instruction kind per dispatch -> clocks per wave64 instruction per
//         SIMD at full occupancy = the issue peak SQ_INSTS_VALU / clock has to be compared with.
//   L1:   every lane gathers 16 bytes from its own random 64-byte line of an 8 KB (L1-resident) or 2 MB (L2-resident) table, or
//         all lanes of a quad / of the wave share a line -> TCP_TOTAL_CACHE_ACCESSES per lane-load and the lane-loads per clock.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/calibrate.hip -o tools/microbench/calibrate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { FMA, FMA_MIX, MAX3, CVT_UBYTE, ALIGNBIT, CNDMASK, MUL_LO_U32, RCP, SQRT, ADD_U32, CNDMASK_S, MUL_F32, MIN_F32,
            PERM, AND_B32, AND_OR, LSHL_OR, LSHLREV, BFE, BFI, CVT_F32_U32, CMP_ADDC, ADD_F32, FMAC, PK_FMA, PK_MUL, MED3, MAX_F32, BCNT, XOR, MAD_U24, CVT_PK_FP8, SUB_F32, LSHL_ADD, ADD3, MOV_SDWA_B2, OR_SDWA, MOV_B32, ADD_F32_SDWA, MUL_F32_SDWA, KINDS };
static const char *kind_name[] = {"v_fma_f32", "v_fma_mix_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_alignbit_b32", "v_cndmask_b32", "v_mul_lo_u32", "v_rcp_f32", "v_sqrt_f32", "v_add_u32", "v_cndmask_b32 (sgpr mask)", "v_mul_f32", "v_min_f32",
    "v_perm_b32", "v_and_b32", "v_and_or_b32", "v_lshl_or_b32", "v_lshlrev_b32", "v_bfe_u32", "v_bfi_b32", "v_cvt_f32_u32", "v_cmp_le_f32 + v_addc_co_u32 (2 insts)", "v_add_f32", "v_fmac_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_med3_f32", "v_max_f32", "v_bcnt_u32_b32", "v_xor_b32", "v_mad_u32_u24", "v_cvt_pk_f32_fp8", "v_sub_f32", "v_lshl_add_u32", "v_add3_u32",
    "v_mov_b32_sdwa dst_sel:BYTE_2 preserve src0_sel:BYTE_1", "v_or_b32_sdwa src0_sel:BYTE_1", "v_mov_b32", "v_add_f32_sdwa src0_sel:WORD_1", "v_mul_f32_sdwa src0_sel:BYTE_1"};

template <int K>
__device__ __forceinline__ void op(float &a, float b, float c) {
    if (K == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(b), "v"(c));
    if (K == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == CVT_UBYTE) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "+v"(a) : "v"(b));
    if (K == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
    if (K == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == RCP) asm volatile("v_rcp_f32 %0, %1" : "+v"(a) : "v"(b));
    if (K == SQRT) asm volatile("v_sqrt_f32 %0, %1" : "+v"(a) : "v"(b));
    if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(0x5555555555555555ull));
    if (K == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(0x07020c0cu));
    if (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
    if (K == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
    if (K == BFE) asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(a));
    if (K == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "+v"(a) : "v"(b));
    if (K == CMP_ADDC) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    if (K == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == CVT_PK_FP8) { float2 t2; asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(t2) : "v"(a)); a = t2.x; }
    if (K == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (K == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
    if (K == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == MOV_SDWA_B2) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(a) : "v"(b));
    if (K == OR_SDWA) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(b), "v"(c));
    if (K == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    if (K == ADD_F32_SDWA) asm volatile("v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a) : "v"(b));
    if (K == MUL_F32_SDWA) asm volatile("v_mul_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a) : "v"(b));
}
// packed f32: two accumulators per instruction (64-bit register pairs)
template <int K>
__device__ __forceinline__ void op2(float2 &a, float2 b, float2 c) {
    if (K == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (K == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
}
template <int K>
__global__ __launch_bounds__(256) void valu_stream2(int iters, float *out) {
    float2 acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = make_float2((float)(threadIdx.x + k), 1.0f);
    const float2 b = make_float2(1.0001f, 0.9999f), c = make_float2(0.5f, 0.25f);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) op2<K>(acc[k], b, c);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += acc[k].x + acc[k].y;
    if (s == 123.456f) out[0] = s;
}

// 16 independent accumulators x 4 rounds = 64 VALU instructions per loop iteration, nothing else vector in the loop
template <int K>
__global__ __launch_bounds__(256) void valu_stream(int iters, float *out) {
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = (float)(threadIdx.x + k);
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 16; k++) op<K>(acc[k], b, c);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += acc[k];
    if (s == 123.456f) out[0] = s;
}

// share = 1: every lane its own line; 4: the lanes of a quad share a line (and an address); 64: the whole wave one address
template <int SHARE>
__global__ __launch_bounds__(256) void gather16(const float *__restrict__ table, uint32_t mask_lines, int iters, float *out) {
    uint32_t s = ((blockIdx.x * 256u + threadIdx.x) / SHARE) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = s * 747796405u + 2891336453u;
            const float4 v = *reinterpret_cast<const float4 *>(table + ((s >> 8) & mask_lines) * 16u);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// one random 64-byte record per lane per step, fetched whole (4 x 16 bytes of one line): the access pattern of an 8-ary node visit whose
// records miss every cache (table far beyond L2 and the 256 MB Infinity Cache) -> what FETCH_SIZE counts per such record, and the rate the
// memory system sustains for it
__global__ __launch_bounds__(256) void gather_record64(const float *__restrict__ table, uint32_t mask_lines, int iters, float *out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        s = s * 747796405u + 2891336453u;
        const float4 *r = reinterpret_cast<const float4 *>(table + (size_t)((s >> 5) & mask_lines) * 16u);
        const float4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += (a.x + b.y) + (c.z + d.w);
    }
    if (acc == 123.456f) out[0] = acc;
}

// the same with 128-byte records (two adjacent 64-byte halves, 128-byte aligned): one fabric request per record would mean a request IS a 128-byte line
__global__ __launch_bounds__(256) void gather_record128(const float *__restrict__ table, uint32_t mask_lines, int iters, float *out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        s = s * 747796405u + 2891336453u;
        const float4 *r = reinterpret_cast<const float4 *>(table + (size_t)((s >> 5) & mask_lines & ~1u) * 16u);
        const float4 a = r[0], b = r[3], c = r[4], d = r[7];
        acc += (a.x + b.y) + (c.z + d.w);
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    float *out;
    CHECK(hipMalloc((void **)&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int cus = 0, clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    const double clk = clock_khz * 1e3;
    printf("# device: %d CUs, %.2f GHz (hipDeviceAttributeClockRate); dispatch index | what | known count | ms | derived\n", cus, clk / 1e9);
    int dispatch = 0;
    const int blocks = cus * 6, iters = 4000;          // 6 waves per SIMD resident, one round: every wave lives for the whole dispatch
    auto time_valu = [&](int kind, auto kernel) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, iters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double wave_insts = (double)blocks * 4 * iters * 64;
            printf("%d | valu %s rep %d | wave_insts %.0f | %.4f ms | %.3f clocks per wave64 instruction per SIMD, %.3f insts per clock per SIMD\n", dispatch++, kind_name[kind], rep,
                   wave_insts, ms, ms * 1e-3 * clk / (wave_insts / (cus * 4.0)), (wave_insts / (cus * 4.0)) / (ms * 1e-3 * clk));
        }
    };
    time_valu(FMA, valu_stream<FMA>);
    time_valu(FMA_MIX, valu_stream<FMA_MIX>);
    time_valu(MAX3, valu_stream<MAX3>);
    time_valu(CVT_UBYTE, valu_stream<CVT_UBYTE>);
    time_valu(ALIGNBIT, valu_stream<ALIGNBIT>);
    time_valu(CNDMASK, valu_stream<CNDMASK>);
    time_valu(MUL_LO_U32, valu_stream<MUL_LO_U32>);
    time_valu(RCP, valu_stream<RCP>);
    time_valu(SQRT, valu_stream<SQRT>);
    time_valu(ADD_U32, valu_stream<ADD_U32>);
    time_valu(CNDMASK_S, valu_stream<CNDMASK_S>);
    time_valu(MUL_F32, valu_stream<MUL_F32>);
    time_valu(MIN_F32, valu_stream<MIN_F32>);
    if (getenv("CALIBRATE_MORE")) {       // instruction classes beyond the ones the roofline uses (developer; not part of the --pmc join)
        time_valu(PERM, valu_stream<PERM>); time_valu(AND_B32, valu_stream<AND_B32>); time_valu(AND_OR, valu_stream<AND_OR>);
        time_valu(LSHL_OR, valu_stream<LSHL_OR>); time_valu(LSHLREV, valu_stream<LSHLREV>); time_valu(BFE, valu_stream<BFE>);
        time_valu(BFI, valu_stream<BFI>); time_valu(CVT_F32_U32, valu_stream<CVT_F32_U32>); time_valu(CMP_ADDC, valu_stream<CMP_ADDC>);
        time_valu(ADD_F32, valu_stream<ADD_F32>); time_valu(FMAC, valu_stream<FMAC>);
        time_valu(PK_FMA, valu_stream2<PK_FMA>); time_valu(PK_MUL, valu_stream2<PK_MUL>);
        time_valu(MED3, valu_stream<MED3>); time_valu(MAX_F32, valu_stream<MAX_F32>); time_valu(BCNT, valu_stream<BCNT>);
        time_valu(XOR, valu_stream<XOR>); time_valu(MAD_U24, valu_stream<MAD_U24>); time_valu(CVT_PK_FP8, valu_stream<CVT_PK_FP8>);
        time_valu(SUB_F32, valu_stream<SUB_F32>); time_valu(LSHL_ADD, valu_stream<LSHL_ADD>); time_valu(ADD3, valu_stream<ADD3>);
        time_valu(MOV_SDWA_B2, valu_stream<MOV_SDWA_B2>); time_valu(OR_SDWA, valu_stream<OR_SDWA>); time_valu(MOV_B32, valu_stream<MOV_B32>);
        time_valu(ADD_F32_SDWA, valu_stream<ADD_F32_SDWA>); time_valu(MUL_F32_SDWA, valu_stream<MUL_F32_SDWA>);
        return 0;
    }

    const int gblocks = cus * 8, giters = 2000;
    for (uint32_t lines : {128u, 32768u}) {               // 8 KB (L1-resident) and 2 MB (L2-resident)
        std::vector<float> h(lines * 16, 1.0f);
        float *d;
        CHECK(hipMalloc((void **)&d, lines * 64));
        CHECK(hipMemcpy(d, h.data(), lines * 64, hipMemcpyHostToDevice));
        auto time_gather = [&](int share, auto kernel) {
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kernel, dim3(gblocks), dim3(256), 0, 0, d, lines - 1, giters, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double lane_loads = (double)gblocks * 256 * giters * 4;
                printf("%d | gather 16 B, table %u KB, %d lanes per line, rep %d | lane_loads %.0f | %.4f ms | %.3f lane-loads per clock per CU, %.3f wave-loads per clock per CU\n", dispatch++,
                       lines * 64 / 1024, share, rep, lane_loads, ms, lane_loads / (ms * 1e-3 * clk) / cus, lane_loads / 64 / (ms * 1e-3 * clk) / cus);
            }
        };
        time_gather(1, gather16<1>);
        time_gather(4, gather16<4>);
        time_gather(64, gather16<64>);
        CHECK(hipFree(d));
    }
    {   // 4 GB of 64-byte records, every fetch a miss all the way to HBM
        const uint32_t lines = 1u << 26;
        float *d;
        CHECK(hipMalloc((void **)&d, (size_t)lines * 64));
        CHECK(hipMemset(d, 0, (size_t)lines * 64));
        const int rblocks = cus * 8, riters = 1000;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather_record64, dim3(rblocks), dim3(256), 0, 0, d, lines - 1, riters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double records = (double)rblocks * 256 * riters;
            printf("%d | random 64-byte records from a 4 GB table (4 x 16 B of one line per lane), rep %d | records %.0f | %.4f ms | %.1f GB/s of records (64 B each), %.1f GB/s if every miss fills a 128-byte line\n",
                   dispatch++, rep, records, ms, records * 64 / (ms * 1e-3) / 1e9, records * 128 / (ms * 1e-3) / 1e9);
        }
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather_record128, dim3(rblocks), dim3(256), 0, 0, d, lines - 1, riters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double records = (double)rblocks * 256 * riters;
            printf("%d | random 128-byte records from a 4 GB table (both 64-byte halves touched), rep %d | records %.0f | %.4f ms | %.1f GB/s of records (128 B each)\n",
                   dispatch++, rep, records, ms, records * 128 / (ms * 1e-3) / 1e9);
        }
        CHECK(hipFree(d));
    }
    return 0;
}
