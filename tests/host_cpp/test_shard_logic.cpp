// test_shard_logic.cpp -- the offset logic of the multi-GPU path (include/x3hip.h: x3_shard_frame_range,
// x3_shard_sample_range, x3_shard_offsets) without a GPU: R host threads play the ranks.  Each encodes the samples
// the library deals it with the CPU oracle (test infrastructure), the "all-gather" is a shared array, and every
// rank copies its sub-stream to the offset the library's scan gives it.  The reassembled buffer must be the
// oracle's encoding of the whole signal, byte for byte -- for worlds of 1..8 ranks, streams shorter than the
// world, tail frames, and non-default frame geometries.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/x3hip.h"
#include "../../oracle/x3_oracle.h"

static int check(uint64_t n, int world, uint32_t block_len, uint32_t bpf) {
  std::vector<int16_t> wav(n);
  x3_synth(2, 0x5A17 + n, 0, n, wav.data());
  x3_params p;
  x3_params_default(&p);
  p.block_len = block_len;
  p.blocks_per_frame = bpf;
  x3o_params op;
  x3o_params_default(&op);
  op.block_len = block_len;
  op.blocks_per_frame = bpf;
  const uint64_t cap = x3_encode_bound(n, &p) + 64;
  std::vector<uint8_t> ref(cap), whole(cap, 0xEE);
  uint64_t ref_len = 0, st[6];
  if (x3o_encode(wav.data(), n, 1, &op, ref.data(), cap, 0, &ref_len, st) != 0) return 1;

  // ranges tile the frames
  const uint64_t F = x3_num_frames(n, &p);
  uint64_t next = 0;
  for (int r = 0; r < world; ++r) {
    uint64_t a, c;
    x3_shard_frame_range(F, r, world, &a, &c);
    if (a != next) return 2;
    next = a + c;
  }
  if (next != F) return 3;

  std::vector<uint64_t> lengths(world, 0);
  std::vector<std::vector<uint8_t>> sub(world);
  std::vector<int> rcs(world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r)
    th.emplace_back([&, r] {
      uint64_t first, cnt;
      x3_shard_sample_range(n, &p, r, world, &first, &cnt);
      if (!cnt) return;
      sub[r].resize(x3_encode_bound(cnt, &p) + 64);
      uint64_t len = 0, s6[6];
      rcs[r] = x3o_encode(wav.data() + first, cnt, 1, &op, sub[r].data(), sub[r].size(), 0, &len, s6);
      lengths[r] = len;  // step 1: the all-gather
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (rcs[r]) return 4;
  std::vector<uint64_t> starts(world + 1);
  x3_shard_offsets(lengths.data(), world, starts.data());
  th.clear();
  for (int r = 0; r < world; ++r)
    th.emplace_back([&, r] {  // step 2: the gather
      if (lengths[r]) std::memcpy(whole.data() + starts[r], sub[r].data(), lengths[r]);
    });
  for (auto& t : th) t.join();
  if (starts[world] != ref_len) return 5;
  for (int r = 0; r <= world; ++r)
    if (starts[r] & 1) return 6;  // sub-streams concatenate without padding
  if (std::memcmp(whole.data(), ref.data(), ref_len) != 0) return 7;
  return 0;
}

int main() {
  int fails = 0;
  const uint64_t sizes[] = {1, 5, 9999, 10000, 10001, 70001, 123457, 400003};
  for (uint64_t n : sizes)
    for (int world = 1; world <= 8; ++world) {
      const int rc = check(n, world, 20, 500);
      if (rc) { std::printf("FAIL n=%llu world=%d rc=%d\n", (unsigned long long)n, world, rc); ++fails; }
    }
  for (int world : {2, 3, 8}) {
    int rc = check(33333, world, 7, 100);
    if (rc) { std::printf("FAIL geometry 7x100 world=%d rc=%d\n", world, rc); ++fails; }
    rc = check(50000, world, 60, 10);
    if (rc) { std::printf("FAIL geometry 60x10 world=%d rc=%d\n", world, rc); ++fails; }
  }
  std::printf(fails ? "test_shard_logic: %d failures\n" : "test_shard_logic: ok\n", fails);
  return fails ? 1 : 0;
}
