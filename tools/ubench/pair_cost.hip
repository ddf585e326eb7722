// pair_cost.hip -- cycles per PAIR OF CODEWORDS of the decoder's parser, the instruction sequences as they stand in
// x3_decode_split_kernel.h, for one wave alone on its SIMD (grid 1024 = one per SIMD) and for two (grid 2048).
//   old:  peek + (ffbh, mad, alignbit, bfe) x 2 + add3 + window update (7) + 3 for the packed index = 19
//   c6 :  the carried-peek sequence (X3S_CHAIN6), 22 instructions, 6 on the chain, hand-interleaved
//   c6n:  the same without the LDS read and its wait
//   ind:  19 independent v_add_u32 (what 19 instructions cost a wave at best)
// Build: hipcc --offload-arch=gfx950 -O3 -o pair_cost pair_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define ITER 2000

template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned long long* out, uint32_t seed) {
  __shared__ uint32_t lds[64 * 32];
  for (int i = 0; i < 32; ++i) lds[threadIdx.x * 32 + i] = (threadIdx.x * 2654435761u + i * 40503u + seed) | 0x01010101u;
  __syncthreads();
  const uint32_t row_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lds[threadIdx.x * 32];
  uint32_t w0 = lds[threadIdx.x * 32], w1 = lds[threadIdx.x * 32 + 1], wn = lds[threadIdx.x * 32 + 2];
  uint32_t s = 5, qb = 0, acc = 0;
  uint32_t zmask = 0xFFFFFFFFu, nwidth = 0u - 2u, fw = 1, lsh = 1, lmul = 2, c124 = 124u;
  uint32_t t = w0, u = w1, b0 = w1, b1 = wn, addr = row_base, pend = 0;
  uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  uint32_t w2 = lds[threadIdx.x * 32 + 3], mp = 0, wna = wn, wnb = wn, pz1 = 1, pz2 = 2, pv1 = 0, pv2 = 1;
  const long long t0 = clock64();
  for (int i = 0; i < ITER; ++i) {
    if (KIND == 0) {
      REP8({
        uint32_t z1, z2, v1, v2, t2, nn1, nn2, tt, m, ad, x1, x2;
        int32_t s2;
        asm volatile(
            "v_alignbit_b32 %[tt], %[w0], %[w1], %[s]\n\t"
            "v_ffbh_u32 %[z1], %[tt]\n\t"
            "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"
            "v_alignbit_b32 %[t2], %[tt], 0, %[nn1]\n\t"
            "v_bfe_u32 %[v1], %[tt], %[nn1], %[fw]\n\t"
            "v_ffbh_u32 %[z2], %[t2]\n\t"
            "v_mad_i32_i24 %[nn2], %[z2], %[zmask], %[nwidth]\n\t"
            "v_bfe_u32 %[v2], %[t2], %[nn2], %[fw]\n\t"
            "v_add3_u32 %[s2], %[s], %[nn1], %[nn2]\n\t"
            "v_ashrrev_i32 %[m], 31, %[s2]\n\t"
            "v_and_b32 %[s], 31, %[s2]\n\t"
            "v_bfi_b32 %[w0], %[m], %[w1], %[w0]\n\t"
            "v_bfi_b32 %[w1], %[m], %[wn], %[w1]\n\t"
            "v_lshl_add_u32 %[qb], %[m], 2, %[qb]\n\t"
            "v_and_or_b32 %[ad], %[qb], %[c124], %[rowb]\n\t"
            "ds_read_b32 %[wn], %[ad]\n\t"
            "v_lshl_add_u32 %[x1], %[z1], %[lsh], %[v1]\n\t"
            "v_lshl_add_u32 %[x2], %[z2], %[lsh], %[v2]\n\t"
            "v_perm_b32 %[x1], %[x2], %[x1], %[sel]\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [z1] "=&v"(z1), [z2] "=&v"(z2), [v1] "=&v"(v1), [v2] "=&v"(v2), [t2] "=&v"(t2), [nn1] "=&v"(nn1),
              [nn2] "=&v"(nn2), [tt] "=&v"(tt), [m] "=&v"(m), [ad] "=&v"(ad), [x1] "=&v"(x1), [x2] "=&v"(x2),
              [s2] "=&v"(s2), [w0] "+v"(w0), [w1] "+v"(w1), [wn] "+v"(wn), [s] "+v"(s), [qb] "+v"(qb)
            : [zmask] "v"(zmask), [nwidth] "v"(nwidth), [fw] "v"(fw), [lsh] "v"(lsh), [c124] "v"(c124),
              [rowb] "v"(row_base), [sel] "v"(0x05040100u));
        acc ^= x1;
      })
    }
    if (KIND == 1 || KIND == 2) {
      REP8({
        uint32_t z1, z2, nn1, t2, m, b2 = 0, nn12, X;
        if (KIND == 1)
          asm volatile(
              "ds_read_b32 %[b2], %[addr]\n\t"
              "v_ffbh_u32 %[z1], %[t]\n\t"
              "v_add_u32 %[s], %[s], %[pend]\n\t"
              "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"
              "v_ashrrev_i32 %[m], 31, %[s]\n\t"
              "v_alignbit_b32 %[t2], %[t], 0, %[nn1]\n\t"
              "v_add_u32 %[nn12], %[nn1], %[nwidth]\n\t"
              "v_ffbh_u32 %[z2], %[t2]\n\t"
              "v_and_b32 %[s], 31, %[s]\n\t"
              "v_mad_i32_i24 %[nn12], %[z2], %[zmask], %[nn12]\n\t"
              "v_bfi_b32 %[b0], %[m], %[b1], %[b0]\n\t"
              "v_bfe_u32 %[X], %[t], %[nn1], %[fw]\n\t"
              "v_lshl_add_u32 %[qb], %[m], 2, %[qb]\n\t"
              "v_bfe_u32 %[t2], %[t], %[nn12], %[fw]\n\t"
              "v_and_or_b32 %[addr], %[qb], %[c124], %[rowb]\n\t"
              "v_mad_u32_u24 %[X], %[z1], %[lmul], %[X]\n\t"
              "s_waitcnt lgkmcnt(0)\n\t"
              "v_bfi_b32 %[b1], %[m], %[b2], %[b1]\n\t"
              "v_mad_u32_u24 %[t2], %[z2], %[lmul], %[t2]\n\t"
              "v_alignbit_b32 %[u], %[b0], %[b1], %[s]\n\t"
              "v_lshl_or_b32 %[X], %[t2], 16, %[X]\n\t"
              "v_alignbit_b32 %[t], %[t], %[u], %[nn12]"
              : [z1] "=&v"(z1), [z2] "=&v"(z2), [nn1] "=&v"(nn1), [t2] "=&v"(t2), [m] "=&v"(m), [b2] "=&v"(b2),
                [nn12] "=&v"(nn12), [X] "=&v"(X), [t] "+v"(t), [u] "+v"(u), [s] "+v"(s), [b0] "+v"(b0),
                [b1] "+v"(b1), [qb] "+v"(qb), [addr] "+v"(addr)
              : [pend] "v"(pend), [zmask] "v"(zmask), [nwidth] "v"(nwidth), [fw] "v"(fw), [lmul] "v"(lmul),
                [c124] "v"(c124), [rowb] "v"(row_base));
        else
          asm volatile(
              "v_ffbh_u32 %[z1], %[t]\n\t"
              "v_add_u32 %[s], %[s], %[pend]\n\t"
              "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"
              "v_ashrrev_i32 %[m], 31, %[s]\n\t"
              "v_alignbit_b32 %[t2], %[t], 0, %[nn1]\n\t"
              "v_add_u32 %[nn12], %[nn1], %[nwidth]\n\t"
              "v_ffbh_u32 %[z2], %[t2]\n\t"
              "v_and_b32 %[s], 31, %[s]\n\t"
              "v_mad_i32_i24 %[nn12], %[z2], %[zmask], %[nn12]\n\t"
              "v_bfi_b32 %[b0], %[m], %[b1], %[b0]\n\t"
              "v_bfe_u32 %[X], %[t], %[nn1], %[fw]\n\t"
              "v_lshl_add_u32 %[qb], %[m], 2, %[qb]\n\t"
              "v_bfe_u32 %[t2], %[t], %[nn12], %[fw]\n\t"
              "v_and_or_b32 %[addr], %[qb], %[c124], %[rowb]\n\t"
              "v_mad_u32_u24 %[X], %[z1], %[lmul], %[X]\n\t"
              "v_bfi_b32 %[b1], %[m], %[b2], %[b1]\n\t"
              "v_mad_u32_u24 %[t2], %[z2], %[lmul], %[t2]\n\t"
              "v_alignbit_b32 %[u], %[b0], %[b1], %[s]\n\t"
              "v_lshl_or_b32 %[X], %[t2], 16, %[X]\n\t"
              "v_alignbit_b32 %[t], %[t], %[u], %[nn12]"
              : [z1] "=&v"(z1), [z2] "=&v"(z2), [nn1] "=&v"(nn1), [t2] "=&v"(t2), [m] "=&v"(m),
                [nn12] "=&v"(nn12), [X] "=&v"(X), [t] "+v"(t), [u] "+v"(u), [s] "+v"(s), [b0] "+v"(b0),
                [b1] "+v"(b1), [qb] "+v"(qb), [addr] "+v"(addr)
              : [pend] "v"(pend), [zmask] "v"(zmask), [nwidth] "v"(nwidth), [fw] "v"(fw), [lmul] "v"(lmul),
                [c124] "v"(c124), [rowb] "v"(row_base), [b2] "v"(b2));
        pend = nn12;
        acc ^= X;
      })
    }
    if (KIND == 5 || KIND == 6) {
      // round 5: the same 19 vector instructions (+1: a third window word) in SOFTWARE-PIPELINED order -- every instruction
      // of the dependent chain (peek, ffbh, mad, shift, ffbh, mad, add3, sign, bfi) is followed by one that does not depend
      // on it: the PREVIOUS pair's packing, this pair's field extraction, the ring address and read one pair ahead (the third
      // window word w2 is brought up to date half a pair late, so the read has a pair and a half to arrive).
      // KIND 6: the thin form (no field extraction, no packing: 15 + read)
#define PAIR5(WNC, WNP, CNT)                                        \
        "v_alignbit_b32 %[tt], %[w0], %[w1], %[s]\n\t"              \
        "v_lshl_add_u32 %[qb], %[mp], 2, %[qb]\n\t"                 \
        "v_ffbh_u32 %[z1], %[tt]\n\t"                               \
        "v_and_or_b32 %[ad], %[qb], %[c124], %[rowb]\n\t"           \
        "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"      \
        "ds_read_b32 %[" WNC "], %[ad]\n\t"                         \
        "v_alignbit_b32 %[t2], %[tt], 0, %[nn1]\n\t"                \
        "v_lshl_add_u32 %[xa], %[pz1], %[lsh], %[pv1]\n\t"          \
        "v_ffbh_u32 %[z2], %[t2]\n\t"                               \
        "v_lshl_add_u32 %[xb], %[pz2], %[lsh], %[pv2]\n\t"          \
        "v_mad_i32_i24 %[nn2], %[z2], %[zmask], %[nwidth]\n\t"      \
        "s_waitcnt lgkmcnt(" CNT ")\n\t"                            \
        "v_bfi_b32 %[w2], %[mp], %[" WNP "], %[w2]\n\t"             \
        "v_add3_u32 %[s2], %[s], %[nn1], %[nn2]\n\t"                \
        "v_bfe_u32 %[pv1], %[tt], %[nn1], %[fw]\n\t"                \
        "v_ashrrev_i32 %[mp], 31, %[s2]\n\t"                        \
        "v_bfe_u32 %[pv2], %[t2], %[nn2], %[fw]\n\t"                \
        "v_bfi_b32 %[w0], %[mp], %[w1], %[w0]\n\t"                  \
        "v_mov_b32 %[pz1], %[z1]\n\t"                               \
        "v_bfi_b32 %[w1], %[mp], %[w2], %[w1]\n\t"                  \
        "v_perm_b32 %[xa], %[xb], %[xa], %[sel]\n\t"                \
        "v_and_b32 %[s], 31, %[s2]\n\t"                             \
        "v_mov_b32 %[pz2], %[z2]\n\t"
#define PAIR6(WNC, WNP, CNT)                                        \
        "v_alignbit_b32 %[tt], %[w0], %[w1], %[s]\n\t"              \
        "v_lshl_add_u32 %[qb], %[mp], 2, %[qb]\n\t"                 \
        "v_ffbh_u32 %[z1], %[tt]\n\t"                               \
        "v_and_or_b32 %[ad], %[qb], %[c124], %[rowb]\n\t"           \
        "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"      \
        "ds_read_b32 %[" WNC "], %[ad]\n\t"                         \
        "v_alignbit_b32 %[t2], %[tt], 0, %[nn1]\n\t"                \
        "s_waitcnt lgkmcnt(" CNT ")\n\t"                            \
        "v_bfi_b32 %[w2], %[mp], %[" WNP "], %[w2]\n\t"             \
        "v_ffbh_u32 %[z2], %[t2]\n\t"                               \
        "v_mov_b32 %[xa], %[tt]\n\t"                                \
        "v_mad_i32_i24 %[nn2], %[z2], %[zmask], %[nwidth]\n\t"      \
        "v_add3_u32 %[s2], %[s], %[nn1], %[nn2]\n\t"                \
        "v_ashrrev_i32 %[mp], 31, %[s2]\n\t"                        \
        "v_and_b32 %[s], 31, %[s2]\n\t"                             \
        "v_bfi_b32 %[w0], %[mp], %[w1], %[w0]\n\t"                  \
        "v_bfi_b32 %[w1], %[mp], %[w2], %[w1]\n\t"
      REP8({
        uint32_t tt, t2, nn1, nn2, ad, xa, xb = 0, z1, z2;
        int32_t s2;
        if (KIND == 5)
          asm volatile(PAIR5("wna", "wnb", "1")
              : [tt] "=&v"(tt), [t2] "=&v"(t2), [nn1] "=&v"(nn1), [nn2] "=&v"(nn2), [ad] "=&v"(ad), [xa] "=&v"(xa), [xb] "=&v"(xb),
                [z1] "=&v"(z1), [z2] "=&v"(z2), [s2] "=&v"(s2), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [s] "+v"(s), [qb] "+v"(qb),
                [mp] "+v"(mp), [wna] "+v"(wna), [wnb] "+v"(wnb), [pz1] "+v"(pz1), [pz2] "+v"(pz2), [pv1] "+v"(pv1), [pv2] "+v"(pv2)
              : [zmask] "v"(zmask), [nwidth] "v"(nwidth), [fw] "v"(fw), [lsh] "v"(lsh), [c124] "v"(c124), [rowb] "v"(row_base), [sel] "v"(0x05040100u));
        else
          asm volatile(PAIR6("wna", "wnb", "1")
              : [tt] "=&v"(tt), [t2] "=&v"(t2), [nn1] "=&v"(nn1), [nn2] "=&v"(nn2), [ad] "=&v"(ad), [xa] "=&v"(xa),
                [z1] "=&v"(z1), [z2] "=&v"(z2), [s2] "=&v"(s2), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [s] "+v"(s), [qb] "+v"(qb),
                [mp] "+v"(mp), [wna] "+v"(wna), [wnb] "+v"(wnb)
              : [zmask] "v"(zmask), [nwidth] "v"(nwidth), [c124] "v"(c124), [rowb] "v"(row_base));
        acc ^= xa;
        { uint32_t tmp = wna; wna = wnb; wnb = tmp; }
      })
    }
    if (KIND == 3) {
      REP8(asm volatile(
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
    }
    if (KIND == 4) {   // 19 three-operand instructions, four independent chains
      REP8(asm volatile(
          "v_bfi_b32 %0, %4, %0, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5\n\t"
          "v_bfi_b32 %0, %4, %0, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5\n\t"
          "v_bfi_b32 %0, %4, %0, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5\n\t"
          "v_bfi_b32 %0, %4, %0, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5\n\t"
          "v_bfi_b32 %0, %4, %0, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(c124));)
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0)
    out[blockIdx.x] = (unsigned long long)(t1 - t0) + ((unsigned long long)(acc + t + u + w0 + w1 + w2 + wna + pz1 + pv1 + pv2 + pz2 + mp + a0 + a1 + a2 + a3) & 1ull);
}

template <int KIND>
static void run(const char* name, int grid) {
  unsigned long long* d;
  (void)hipMalloc(&d, grid * 8);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 1u);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 2u);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-44s waves=%5d  cycles/pair = %.1f\n", name, grid, s / grid / ((double)ITER * 8));
  (void)hipFree(d);
}

int main() {
  for (int grid : {1024, 2048, 3072}) {
    run<0>("old pair (19 + wait)", grid);
    run<1>("carried peek (22, LDS read inside)", grid);
    run<2>("carried peek without the LDS read (20)", grid);
    run<3>("19 v_add_u32, four chains", grid);
    run<4>("19 three-operand ops, four chains", grid);
    run<5>("software-pipelined pair (19 + 3 moves, read 1.5 pairs ahead)", grid);
    run<6>("software-pipelined THIN pair (15 + read)", grid);
  }
  return 0;
}
