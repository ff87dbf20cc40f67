// A C++ caller of the multi-GPU exchange with the collective issued by the library itself — the shape of the reference's own driver,
// linemodLevelup/test.cpp:111-130 (train, then Detector::match per frame), one process per GPU, no Python.  Run with world size 1 (what a
// one-GPU box allows): rank 0 makes the RCCL id, creates its communicator, and every frame goes submit -> lm_detector_exchange_group
// (pack + ncclAllGather on the exchange stream + merge) -> collect; the result must equal lm_detector_match on the same frame, record by
// record.  With more GPUs the same program runs once per rank (MPI / a launcher carries the 128-byte id).
//   hipcc -O2 -I include tests/cpp/exchange_rccl_smoke.cpp -L 6dpose_amd -lamdlinemod -Wl,-rpath,$PWD/6dpose_amd -o /tmp/exchange_rccl_smoke
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "amd_linemod.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lm_last_error()); return 1; } \
    } while (0)

int main() {
    if (lm_device_count() <= 0) { fprintf(stderr, "no GPU\n"); return 2; }
    if (!lm_comm_available()) { fprintf(stderr, "librccl.so not found\n"); return 3; }
    const int W = 640, H = 480, T[2] = {4, 8};
    // a frame with structure: blocks of random colour on a depth staircase
    std::vector<uint8_t> rgb((size_t)W * H * 3);
    std::vector<uint16_t> dep((size_t)W * H);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    uint8_t col[20][15][3];
    for (auto& row : col) for (auto& c : row) for (int k = 0; k < 3; ++k) c[k] = (uint8_t)(rnd() & 255);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int bx = x / 32, by = y / 32;
            for (int k = 0; k < 3; ++k) rgb[((size_t)y * W + x) * 3 + k] = (uint8_t)(col[bx][by][k] + ((x * 7 + y * 3 + k) & 7));
            dep[(size_t)y * W + x] = (uint16_t)(700 + 40 * ((bx + 2 * by) % 7) + (x % 32) / 2 + (y % 32) / 3);
        }
    lm_detector* d = nullptr;
    CHECK(lm_detector_create(63, T, 2, 0, &d));
    int trained = 0;
    for (int i = 0; i < 12; ++i) {                              // templates cut out of the frame itself: they match where they came from
        std::vector<uint8_t> mask((size_t)W * H, 0);
        const int x0 = 40 + 45 * i, y0 = 60 + 25 * (i % 5), w = 90 + 4 * i, h = 100;
        for (int y = y0; y < y0 + h && y < H; ++y) memset(&mask[(size_t)y * W + x0], 255, (size_t)(x0 + w < W ? w : W - x0));
        const int id = lm_detector_add_template(d, rgb.data(), dep.data(), mask.data(), W, H, "obj");
        if (id < -1) { fprintf(stderr, "add_template: %s\n", lm_last_error()); return 1; }
        trained += id >= 0;
    }
    if (trained < 4) { fprintf(stderr, "only %d templates could be extracted\n", trained); return 1; }
    lm_match* want = nullptr;
    size_t n_want = 0;
    CHECK(lm_detector_match(d, rgb.data(), dep.data(), W, H, 70.f, nullptr, 0, nullptr, &want, &n_want));

    unsigned char id[128];
    CHECK(lm_comm_unique_id(id));
    lm_comm* comm = nullptr;
    CHECK(lm_comm_create(id, 0, 1, 0, &comm));
    CHECK(lm_detector_set_shard(d, lm_comm_rank(comm), lm_comm_world(comm)));
    const int capacity = 4096, group = 3;
    const size_t block = lm_exchange_block_bytes(capacity);
    void *send = nullptr, *recv = nullptr;
    if (hipMalloc(&send, block * group) != hipSuccess || hipMalloc(&recv, block * group * lm_comm_world(comm)) != hipSuccess) return 1;
    CHECK(lm_detector_set_frame(d, rgb.data(), dep.data(), W, H, nullptr));
    for (int round = 0; round < 2; ++round) {                    // a group of three frames in flight, twice
        const uint64_t first = lm_detector_frames_submitted(d);
        for (int k = 0; k < group; ++k) CHECK(lm_detector_submit(d, 70.f, nullptr, 0));
        CHECK(lm_detector_exchange_group(d, comm, first, group, send, recv, capacity));
        for (int k = 0; k < group; ++k) {
            lm_match* got = nullptr;
            size_t n_got = 0;
            int failed = 0;
            CHECK(lm_detector_exchange_collect(d, &got, &n_got, &failed));
            if (failed || n_got != n_want || memcmp(got, want, n_got * sizeof(lm_match)) != 0) {
                fprintf(stderr, "frame %d of round %d: failed %d, %zu records against %zu\n", k, round, failed, n_got, n_want);
                return 1;
            }
            lm_free(got);
        }
    }
    printf("ok: %d templates, %zu records per frame through lm_detector_exchange_group (RCCL world %d)\n", trained, n_want, lm_comm_world(comm));
    lm_free(want);
    (void)hipFree(send); (void)hipFree(recv);
    lm_comm_destroy(comm);
    lm_detector_destroy(d);
    return n_want > 0 ? 0 : 1;
}
