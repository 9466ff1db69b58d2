// Host-side internals of libmil.so shared by mil_api.cpp (single-GPU entry points) and mil_mg.cpp (multi-GPU strips /
// frames + gather).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mil.h"
#include "mil_params.h"

namespace milhost {

constexpr int RING = 4;
constexpr int SLOT_STREAMS = 4;
constexpr uint32_t LUT11_N = 0x3C1, LUT10_N = 0x1E1;

// One staged copy of a call's light arrays.  Layout of `host` and `dev` (dir_b = MIL_MAX_DIR_LIGHTS * 32,
// pt_b = capacity * 32): dir_raw | dir_prep | point_raw | point_prep (n x 32 bytes each) | point_cull (n x 16).
struct Slot {
    uint8_t *host = nullptr; // pinned
    uint8_t *dev = nullptr;
    // The last kernel that read this slot, PER STREAM: every call names its own stream, and a slot may be recycled only
    // when its readers on all of them are done.
    hipEvent_t done[SLOT_STREAMS] = {};
    hipStream_t done_stream[SLOT_STREAMS] = {};
    bool done_used[SLOT_STREAMS] = {};
    hipEvent_t staged = nullptr; // upload of this slot's content finished (recorded on the copy stream)
    bool valid = false;          // holds staged content
    uint32_t n_dir = 0, n_point = 0;
    uint64_t gen = 0;            // bumped whenever new content is staged into this slot
};

// Workspace of the two-kernel tiled path (k_bin -> k_order -> k_shade, mil_device.hpp): counters, the launch order and per
// half tile {list length, list offset, rank} - sized by the frame (16 bytes per half tile) - and the POOL the compact light lists
// live in - sized by how many (half tile, light) pairs a frame has, not by the frame.  mil_create reserves it for a
// 7680 x 4320 frame and 3072 list entries per point light of the context's capacity (4096 lights: 24 MiB; the 8K / 4096-light
// frame of BASELINE.json's cfg4 asks for 7.65 M entries): 29 MB where rounds 1-2 reserved 133 MB for 4K and 531 MB for 8K.  No
// per-frame call allocates for frames up to the reserved size (mil_reserve raises it ahead of time); a larger frame grows
// the workspace inside the call once (counted in frame_call_allocs), as the reference recreates its attachments when the
// window is resized (render/pbr/mod.rs:263-346).  A pool too small for a frame's lists costs speed, never correctness:
// the macro tiles that found no room are shaded by the brute-force loop.
struct TileWorkspace {
    uint8_t *base = nullptr;            // one allocation
    uint32_t *hist = nullptr, *total = nullptr, *pool_used = nullptr, *order = nullptr, *rank = nullptr;
    uint2 *meta = nullptr;
    uint16_t *pool = nullptr;
    uint32_t items_cap = 0, pool_cap = 0, list_cap = 0;
    size_t bytes = 0;
    hipEvent_t done = nullptr;          // the last launch that used it ...
    hipStream_t last_stream = nullptr;  // ... and its stream: a launch on another stream waits for it (one workspace per context)
    bool used = false;
    bool pinned = false;                // a launch that may have been captured into a HIP graph holds these pointers: never freed again
    uint32_t frame_call_allocs = 0;     // times a per-frame call had to grow it
};
constexpr size_t WS_COUNTER_BYTES = (size_t)MIL_WS_BUCKETS * MIL_WS_SHARDS * 4 + 128 + (MIL_WS_POOL_PARTS + 1) * 128;   // hist; total; pool_used per partition + brute count, 128 bytes apart

constexpr int ROW_ORDER_CACHE = 16;
struct RowOrderEntry {
    uint16_t order[MIL_MAX_ROW_ORDER];
    uint8_t key[192];
    bool valid = false;
};

// Light arrays that already lie in device memory in the slot layout (prepared there by a kernel on the caller's stream):
// nothing is staged, recorded or waited for.
struct PreStaged {
    const uint8_t *dev;
    uint32_t capacity;           // point-light capacity the layout was cut for
};

} // namespace milhost

struct mil_ctx {
    int device = 0;
    uint32_t max_point = 0;
    size_t slot_bytes = 0;
    milhost::Slot ring[milhost::RING];
    int next = 0, cur = -1;      // cur: slot holding the most recently used content
    uint64_t gen = 0;
    uint8_t *scratch = nullptr;  // host: the content of this call, built before it is compared with the ring
    hipStream_t copy_stream = nullptr;
    uint8_t *lut = nullptr;      // device: lut11 then lut10
    uint32_t *tile_counts = nullptr, *tile_stamps = nullptr;
    uint32_t tile_counts_cap = 0, tile_stamps_cap = 0;   // tiles the diagnostic buffers hold
    uint32_t tune = 0;
    milhost::RowOrderEntry row_cache[milhost::ROW_ORDER_CACHE];   // recent K2 row orders and what each was computed from (order_rows)
    int row_cache_next = 0;
    uint32_t tonemap_rows = 0;   // K4 rows per lane            } tuning knobs: read from the environment only when
    uint32_t fxaa_rows = 2;      // K5 output rows per lane      } MIL_DEBUG=1 is set (mil_create); the defaults are what
    uint32_t k1_rows = 1;        // K1 pixels per lane           } the parity suite covers
    milhost::TileWorkspace ws;
    uint32_t two_kernel_min = MIL_TWO_KERNEL_DEFAULT;   // point lights from which the k_bin -> k_shade form is used (fast mode)
    uint8_t *static_scratch = nullptr;   // this device's copy of mil_noalloc.hip's __device__ array (hipGetSymbolAddress)
    int last_hip = 0;
};

namespace milhost {

int hip_fail(mil_ctx *c, hipError_t e);
int set_device(mil_ctx *c);

// mil_light_ex / mil_shade_frame / mil_shade_bands: one routine.
//   bands       NULL = the planes are a strip of contiguous rows
//   out_frame   with bands: `out` is the assembled frame (rows of the whole frame) instead of the compacted local rows
//   pre         NULL = stage the host light arrays through the ring; else they are already on the device
//               (ex->point_lights is then a DEVICE pointer and is not read on the host)
// shade_check is the argument validation alone (no side effect); shade runs it first.
int shade_check(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame, const mil_directional_light *lights,
                const mil_push_constants *push, const mil_extras *ex, const mil_frame_passes *passes, void *out,
                uint32_t out_pitch_px, mil_out_format fmt, const mil_band_layout *bands, bool out_frame, const PreStaged *pre);
int shade(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame, const mil_directional_light *lights,
          const mil_push_constants *push, const mil_extras *ex, const mil_frame_passes *passes, void *out,
          uint32_t out_pitch_px, mil_out_format fmt, uint32_t flags, void *stream, const mil_band_layout *bands,
          bool out_frame, const PreStaged *pre);

// mil_noalloc.hip (the one HIP translation unit of the host layer): a kernel that idles a stream for a while (test hook)
hipError_t launch_delay(hipStream_t s, uint32_t microseconds);

// mil_noalloc.hip: per context at mil_create, with its device current: hipGetSymbolAddress of the static scratch array
int noalloc_init(mil_ctx *c);

} // namespace milhost
