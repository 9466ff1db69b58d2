// Host runtime of libmil.so: the C ABI declared in include/mil.h.
// Owns a small ring of staged light buffers (the counterpart of the reference's
// per-frame CpuBufferPool ring, render/pools.rs:42-57); everything else belongs to
// the caller.  The per-frame calls allocate nothing after mil_create; the external-memory
// set-up calls are the only other allocators.
#include <hip/hip_runtime.h>
#include <math.h>
#include <new>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "mil_internal.h"

static_assert(sizeof(mil_frame_matrix_data) == 272, "FrameMatrixData is 272 bytes (render/ubo.rs:28-41)");
static_assert(sizeof(mil_directional_light) == 32, "DirectionalLight std140 stride is 32 (render/ubo.rs:54-63)");
static_assert(sizeof(mil_point_light) == 32, "point light shares the 32-byte stride");
static_assert(sizeof(mil_hosek_wilkie_params) == 176, "HosekWilkieParams is 11 vec3 at 16-byte stride (render/hosek/mod.rs:222-256)");
static_assert(sizeof(mil_push_constants) == 12, "push constants are 12 bytes (fs_deferred_lighting.glsl:24-27)");

using namespace milhost;

struct mil_external_memory {
    bool imported = false;
    size_t size = 0;
    hipExternalMemory_t ext = nullptr;             // imported: the external-memory object and its mapped buffer
    void *mapped = nullptr;
    hipMemGenericAllocationHandle_t handle = {};   // exportable: physical allocation + the virtual range it is mapped at
    void *va = nullptr;
    bool have_handle = false, is_mapped = false;
};

namespace milhost {

int hip_fail(mil_ctx *c, hipError_t e)
{
    if (c) c->last_hip = (int)e;
    return MIL_ERR_HIP;
}

int set_device(mil_ctx *c)
{
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != c->device) e = hipSetDevice(c->device);
    return e == hipSuccess ? MIL_OK : hip_fail(c, e);
}

} // namespace milhost

#define HIP_TRY(c, expr)                                    \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return hip_fail((c), e_);     \
    } while (0)

namespace {

double ufloat_dec(uint32_t code, int mbits)
{
    uint32_t e = code >> mbits, m = code & ((1u << mbits) - 1u);
    if (e == 0) return ldexp((double)m, -(14 + mbits));
    return ldexp(1.0 + (double)m / (double)(1u << mbits), (int)e - 15);
}
uint8_t srgb8(double c)
{
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    double s = c <= 0.0031308 ? 12.92 * c : 1.055 * pow(c, 1.0 / 2.4) - 0.055;
    return (uint8_t)floor(s * 255.0 + 0.5);
}

bool bands_ok(const mil_band_layout *b)
{
    return b->band_rows >= MIL_TILE && (b->band_rows & (b->band_rows - 1u)) == 0 && b->band_rows <= (1u << 20) &&
           b->n_ranks >= 1 && b->n_ranks <= 4096 && b->rank < b->n_ranks && b->first_band < (1u << 20);
}
// frame row of local row k (bands == NULL: a contiguous strip starting at y_offset)
uint64_t frame_row(const mil_band_layout *b, uint32_t y_offset, uint32_t k)
{
    if (!b) return (uint64_t)y_offset + k;
    return (((uint64_t)(k / b->band_rows) + b->first_band) * b->n_ranks + b->rank) * b->band_rows + k % b->band_rows;
}

int check_gbuffer(const mil_gbuffer *gb, const mil_push_constants *push, const mil_band_layout *bands)
{
    if (!gb || !push) return MIL_ERR_BAD_ARG;
    if (!gb->normal_l_model || !gb->albedo_occlusion || !gb->roughness_metallic || !gb->depth) return MIL_ERR_BAD_ARG;
    if (gb->width == 0 || gb->height == 0 || gb->pitch_px < gb->width) return MIL_ERR_BAD_ARG;
    if ((uint64_t)gb->pitch_px * gb->height >= (1ull << 30)) return MIL_ERR_BAD_ARG;   // the kernels use 32-bit texel offsets (1 Gpx per plane)
    if (!(push->resolution[0] >= 1.0f) || !(push->resolution[1] >= 1.0f)) return MIL_ERR_BAD_ARG;
    if ((float)gb->width > push->resolution[0]) return MIL_ERR_BAD_ARG;
    if (bands && (!bands_ok(bands) || gb->y_offset != 0)) return MIL_ERR_BAD_ARG;
    if ((double)frame_row(bands, gb->y_offset, gb->height - 1) + 1.0 > (double)push->resolution[1]) return MIL_ERR_BAD_ARG;
    return MIL_OK;
}

void slot_pointers(mil::ShadeParams &p, const uint8_t *dev, uint32_t capacity)
{
    const size_t dir_b = (size_t)MIL_MAX_DIR_LIGHTS * 32, pt_b = (size_t)capacity * 32;
    p.dir_raw = (const float4 *)dev;
    p.dir_prep = (const float4 *)(dev + dir_b);
    p.point_raw = (const float4 *)(dev + 2 * dir_b);
    p.point_prep = (const float4 *)(dev + 2 * dir_b + pt_b);
    p.point_cull = (const float4 *)(dev + 2 * dir_b + 2 * pt_b);
}

// the first MIL_DIR_INLINE directional records travel in the kernarg, in the launch's mode layout
void inline_dir_records(mil::ShadeParams &p, const mil_directional_light *dl, uint32_t n_dir, bool strict)
{
    for (uint32_t i = 0; i < n_dir && i < MIL_DIR_INLINE; i++) {
        const mil_directional_light &l = dl[i];
        float *q = p.dir_inline[i];
        q[0] = l.direction[0]; q[1] = l.direction[1]; q[2] = l.direction[2];
        if (strict) {
            q[3] = l.intensity; q[4] = l.color[0]; q[5] = l.color[1]; q[6] = l.color[2];
        } else {
            q[3] = 0.0f; q[4] = l.color[0] * l.intensity; q[5] = l.color[1] * l.intensity; q[6] = l.color[2] * l.intensity;
        }
        q[7] = 0.0f;
    }
}

// does the slot hold exactly these light arrays?  (raw copies; the prepared arrays derive from them)
bool slot_holds(const mil_ctx *c, const Slot &sl, const mil_directional_light *dl, uint32_t n_dir, const mil_point_light *pl, uint32_t n_point)
{
    if (!sl.valid || sl.n_dir != n_dir || sl.n_point != n_point) return false;
    const size_t dir_b = (size_t)MIL_MAX_DIR_LIGHTS * 32;
    if (n_point && memcmp(sl.host + 2 * dir_b, pl, (size_t)n_point * 32)) return false;
    for (uint32_t i = 0; i < n_dir; i++)
        if (memcmp(sl.host + 32 * i, &dl[i], 28)) return false;      // the pad word is not part of the light
    return true;
}

int slot_wait_readers(mil_ctx *c, Slot &sl)
{
    for (int i = 0; i < SLOT_STREAMS; i++)
        if (sl.done_used[i]) {
            HIP_TRY(c, hipEventSynchronize(sl.done[i]));
            sl.done_used[i] = false;
        }
    return MIL_OK;
}

// Stage the light arrays and fill the light part of ShadeParams.  Static lights (the common case frame to frame) are
// recognised by comparing the inputs with the raw copies the ring slots hold - the current slot first, then the others,
// so a call with other content in between (mil_count_pairs, mil_transparency_accumulate) does not evict the frame's
// lights - and then nothing is rebuilt or uploaded.  Otherwise the next ring slot is filled and uploaded on the
// context's copy stream, so the upload overlaps whatever the caller's stream is still running, and the caller's stream
// waits on the slot's `staged` event.
int stage_lights(mil_ctx *c, const mil_directional_light *dl, uint32_t n_dir, const mil_point_light *pl,
                 uint32_t n_point, bool strict, hipStream_t s, mil::ShadeParams &p, Slot *&slot_out)
{
    const size_t dir_b = (size_t)MIL_MAX_DIR_LIGHTS * 32, pt_b = (size_t)c->max_point * 32;
    inline_dir_records(p, dl, n_dir, strict);
    p.n_dir = n_dir;
    p.n_point = n_point;
    slot_out = nullptr;
    // a device copy is needed only beyond the inline records, or for point lights
    if (n_dir <= MIL_DIR_INLINE && n_point == 0) return MIL_OK;
    Slot *sl = nullptr;
    if (c->cur >= 0 && slot_holds(c, c->ring[c->cur], dl, n_dir, pl, n_point)) sl = &c->ring[c->cur];
    for (int i = 0; !sl && i < RING; i++)
        if (i != c->cur && slot_holds(c, c->ring[i], dl, n_dir, pl, n_point)) { sl = &c->ring[i]; c->cur = i; }
    if (!sl) {
        if (c->next == c->cur) c->next = (c->next + 1) % RING;      // keep the slot in use; take the one after it
        sl = &c->ring[c->next];
        int rc = slot_wait_readers(c, *sl);                         // kernels of RING contents ago may still read it
        if (rc) return rc;
        sl->valid = false;
        float *h_dir_raw = (float *)sl->host, *h_dir_prep = (float *)(sl->host + dir_b);
        float *h_pt_raw = (float *)(sl->host + 2 * dir_b), *h_pt_prep = (float *)(sl->host + 2 * dir_b + pt_b);
        float *h_pt_cull = (float *)(sl->host + 2 * dir_b + 2 * pt_b);
        for (uint32_t i = 0; i < n_dir; i++) {
            const mil_directional_light &l = dl[i];
            float *r = h_dir_raw + 8 * i, *q = h_dir_prep + 8 * i;
            r[0] = l.direction[0]; r[1] = l.direction[1]; r[2] = l.direction[2]; r[3] = l.intensity;
            r[4] = l.color[0]; r[5] = l.color[1]; r[6] = l.color[2]; r[7] = 0.0f;
            q[0] = l.direction[0]; q[1] = l.direction[1]; q[2] = l.direction[2]; q[3] = 0.0f;
            q[4] = l.color[0] * l.intensity; q[5] = l.color[1] * l.intensity; q[6] = l.color[2] * l.intensity; q[7] = 0.0f;
        }
        for (uint32_t i = 0; i < n_point; i++) {
            const mil_point_light &l = pl[i];
            float *r = h_pt_raw + 8 * i, *q = h_pt_prep + 8 * i;
            memcpy(r, &l, 32);
            const float r2 = l.radius * l.radius;
            q[0] = l.position[0]; q[1] = l.position[1]; q[2] = l.position[2]; q[3] = r2;
            q[4] = l.color[0] * l.intensity; q[5] = l.color[1] * l.intensity; q[6] = l.color[2] * l.intensity;
            q[7] = 1.0f / r2;
            float *k = h_pt_cull + 4 * i;
            k[0] = l.position[0]; k[1] = l.position[1]; k[2] = l.position[2]; k[3] = l.radius;
        }
        const size_t nd = (size_t)n_dir * 32, np = (size_t)n_point * 32;
        if (nd) HIP_TRY(c, hipMemcpyAsync(sl->dev, sl->host, dir_b + nd, hipMemcpyHostToDevice, c->copy_stream));
        if (np) HIP_TRY(c, hipMemcpyAsync(sl->dev + 2 * dir_b, sl->host + 2 * dir_b, 2 * pt_b + np / 2, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(sl->staged, c->copy_stream));
        sl->n_dir = n_dir; sl->n_point = n_point;
        sl->gen = ++c->gen;
        sl->valid = true;
        c->cur = c->next;
        c->next = (c->next + 1) % RING;
    }
    HIP_TRY(c, hipStreamWaitEvent(s, sl->staged, 0));
    slot_pointers(p, sl->dev, c->max_point);
    slot_out = sl;
    return MIL_OK;
}

// fast path: world.xyz - cam = ((inv_view - [0|cam]) * (v / v.w)).xyz with v = inv_proj * clip and
// clip = (x*sx+ox, y*sy+oy, depth, 1): fold everything acting on (x, y_global, depth, 1) into one 4x4, in double
void fold_position_matrix(const mil_frame_matrix_data *f, float res_x, float res_y, float *out16)
{
    const float *ip = f->inv_projection, *iv = f->inv_view;   // column-major: m[c*4 + r]
    double C[4][4] = { { 2.0 / res_x, 0, 0, 1.0 / res_x - 1.0 }, { 0, 2.0 / res_y, 0, 1.0 / res_y - 1.0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    double PC[4][4], A[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double a = 0;
            for (int k = 0; k < 4; k++) a += (double)ip[k * 4 + r] * C[k][c];
            PC[r][c] = a;
        }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            double a = 0;
            for (int k = 0; k < 4; k++) a += ((double)iv[k * 4 + r] - (k == 3 ? (double)f->camera_position[r] : 0.0)) * PC[k][c];
            A[r][c] = a;   // rows 0-2 yield P - cameraPosition
        }
    for (int c = 0; c < 4; c++) A[3][c] = PC[3][c];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out16[r * 4 + c] = (float)A[r][c];
}

void fill_common(mil::ShadeParams &p, const mil_gbuffer *gb, const mil_frame_matrix_data *f, const mil_push_constants *push,
                 const mil_band_layout *bands)
{
    memset(&p, 0, sizeof(p));
    p.normal = gb->normal_l_model; p.albedo = gb->albedo_occlusion; p.rm = gb->roughness_metallic; p.depth = gb->depth;
    p.width = gb->width; p.height = gb->height; p.pitch = gb->pitch_px; p.y_offset = gb->y_offset;
    if (bands) {
        uint32_t sh = 0;
        while ((1u << sh) < bands->band_rows) sh++;
        p.band_shift = sh;
        p.band_extra = (bands->n_ranks - 1u) * bands->band_rows;     // one rank: 0, the dealing is the identity
        p.y_offset = (bands->first_band * bands->n_ranks + bands->rank) * bands->band_rows;
    }
    memcpy(p.inv_proj, f->inv_projection, 64);
    memcpy(p.inv_view, f->inv_view, 64);
    memcpy(p.cam, f->camera_position, 12);
    p.res_x = push->resolution[0]; p.res_y = push->resolution[1];
    p.sx = 2.0f / p.res_x; p.ox = 1.0f / p.res_x - 1.0f;
    p.sy = 2.0f / p.res_y; p.oy = 1.0f / p.res_y - 1.0f;
    fold_position_matrix(f, p.res_x, p.res_y, p.pos_m);
    p.tiles_x = (gb->width + MIL_TILE - 1) / MIL_TILE;
    p.tiles_y = (gb->height + MIL_TILE - 1) / MIL_TILE;
    for (uint32_t i = 0; i < MIL_MAX_ROW_ORDER; i++) p.row_order[i] = (uint16_t)i;
}

// Scheduling weight of the local tile rows of a launch: w[i] = number of point lights whose bounding sphere can project
// onto local tile row i.  Lights behind the camera weigh nothing; a sphere the near plane cuts counts for every row.
// O(n_point + tile rows): every light adds +1 / -1 at the ends of its range of FRAME tile rows, a prefix sum gives the
// frame's weights, and the launch's local rows pick theirs.
void row_weights(const mil_frame_matrix_data *f, float res_y, uint32_t height, uint32_t y_offset, const mil_band_layout *bands,
                 const mil_point_light *pl, uint32_t n_point, uint32_t *w)
{
    const uint32_t R = (height + MIL_TILE - 1) / MIL_TILE;
    for (uint32_t i = 0; i < R; i++) w[i] = 0;
    if (!(res_y >= 1.0f) || res_y > 16777216.0f) return;
    // frame rows are counted from the launch's first frame row onwards in units of 16 rows: a strip may start on any row,
    // so its local tile row i covers frame rows y_offset + 16 i .. + 15
    const double base = bands ? 0.0 : (double)y_offset;
    const uint32_t FR = (uint32_t)(((double)res_y - base + MIL_TILE - 1) / MIL_TILE) + 1;     // tile rows from `base` to the frame's end
    if (FR > (1u << 21)) return;
    static thread_local std::vector<int32_t> diff;
    diff.assign(FR + 1, 0);
    const float *V = f->view, *P = f->projection;      // column-major
    for (uint32_t i = 0; i < n_point; i++) {
        const float *c = pl[i].position;
        const double r = pl[i].radius;
        if (!(r > 0.0)) continue;                       // radius 0 (or NaN): reaches no pixel
        const double vy = (double)V[1] * c[0] + (double)V[5] * c[1] + (double)V[9] * c[2] + V[13];
        const double vz = (double)V[2] * c[0] + (double)V[6] * c[1] + (double)V[10] * c[2] + V[14];
        const double vw = (double)V[3] * c[0] + (double)V[7] * c[1] + (double)V[11] * c[2] + V[15];
        double lo = -1.0, hi = 1.0;                     // NDC y range; whole frame unless the sphere is safely in front
        if (P[11] != 0.0f) {
            // clip w = P[11] z + P[15] w at the sphere's near / far side (P[11] = -1 for a perspective, view z < 0 in front)
            const double wn = (double)P[11] * (vz + r) + (double)P[15] * vw, wf = (double)P[11] * (vz - r) + (double)P[15] * vw;
            if (wn <= 0.0 && wf <= 0.0) continue;       // wholly behind the camera
            if (wn > 1e-6 && wf > 1e-6) {
                const double a = (double)P[5] * (vy - r) + (double)P[13] * vw, b = (double)P[5] * (vy + r) + (double)P[13] * vw;
                const double c0 = a / wn, c1 = a / wf, c2 = b / wn, c3 = b / wf;
                lo = c0; hi = c0;
                for (double v : { c1, c2, c3 }) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            }
        }
        if (!(lo == lo) || !(hi == hi)) { lo = -1.0; hi = 1.0; }
        const double y0 = (lo * 0.5 + 0.5) * res_y - base, y1 = (hi * 0.5 + 0.5) * res_y - base;      // rows from `base`
        if (y1 < 0.0 || y0 >= (double)res_y - base) continue;
        const uint32_t g0 = y0 <= 0.0 ? 0u : (uint32_t)(y0 / MIL_TILE);
        const uint32_t g1 = y1 >= (double)res_y - base ? FR - 1 : (uint32_t)(y1 / MIL_TILE);
        diff[g0] += 1;
        diff[(g1 < FR ? g1 : FR - 1) + 1] -= 1;
    }
    for (uint32_t g = 1; g <= FR; g++) diff[g] += diff[g - 1];
    if (!bands) {
        for (uint32_t k = 0; k < R && k < FR; k++) w[k] = (uint32_t)diff[k];
    } else {
        const uint32_t tpb = bands->band_rows / MIL_TILE;   // tile rows per band
        for (uint32_t k = 0; k < R; k++) {                  // local tile row k lies in the rank's band k / tpb + first_band
            const uint64_t g = (((uint64_t)(k / tpb) + bands->first_band) * bands->n_ranks + bands->rank) * tpb + k % tpb;
            if (g < FR) w[k] = (uint32_t)diff[g];
        }
    }
}

// Scheduling heuristic of the tiled kernel (speed only; any permutation is correct): tile rows are launched
// in descending order of the number of point lights whose bounding sphere can project onto them, ties from the
// bottom of the frame upwards, so that the expensive rows are in flight early and the launch ends on cheap ones.
struct RowOrderKey { float view[16], proj[16]; uint32_t height, y_offset, n_point, band_rows, n_ranks, rank, first_band; float res_y; uint64_t lights_gen; };
void order_rows(mil_ctx *c, const Slot *lights, mil::ShadeParams &p, const mil_frame_matrix_data *f, const mil_band_layout *bands,
                uint32_t gb_y_offset, const mil_point_light *pl, uint32_t n_point)
{
    const uint32_t R = p.tiles_y;
    if (R > MIL_MAX_ROW_ORDER || !pl) return;
    // same lights (slot generation), camera and strip as the last call: same order
    RowOrderKey key;
    memset(&key, 0, sizeof(key));
    memcpy(key.view, f->view, 64); memcpy(key.proj, f->projection, 64);
    key.height = p.height; key.y_offset = gb_y_offset; key.n_point = n_point; key.res_y = p.res_y; key.lights_gen = lights ? lights->gen : 0;
    if (bands) { key.band_rows = bands->band_rows; key.n_ranks = bands->n_ranks; key.rank = bands->rank; key.first_band = bands->first_band; }
    static_assert(sizeof(c->row_cache[0].key) >= sizeof(RowOrderKey), "row order key storage");
    // a frame is often shaded in several launches (chunks of a rank's rows, strips): a few orders are kept side by side
    for (int e = 0; e < ROW_ORDER_CACHE; e++)
        if (c->row_cache[e].valid && !memcmp(c->row_cache[e].key, &key, sizeof(key))) {
            memcpy(p.row_order, c->row_cache[e].order, sizeof(p.row_order));
            return;
        }
    uint32_t w[MIL_MAX_ROW_ORDER];
    row_weights(f, p.res_y, p.height, gb_y_offset, bands, pl, n_point, w);
    // stable counting order: weight descending, then row index descending (bottom first)
    uint32_t n = 0;
    uint32_t wmax = 0;
    for (uint32_t i = 0; i < R; i++) wmax = w[i] > wmax ? w[i] : wmax;
    for (uint32_t lvl = wmax + 1; lvl-- > 0;)
        for (uint32_t i = R; i-- > 0;)
            if (w[i] == lvl) p.row_order[n++] = (uint16_t)i;
    if (c->tune & 0x70000u) {          // EXPERIMENT (MIL_DEBUG=1 MIL_TUNE): launch orders that mix cheap and expensive rows
        uint16_t s0[MIL_MAX_ROW_ORDER];
        memcpy(s0, p.row_order, sizeof(s0));
        if (c->tune & 0x10000u) for (uint32_t i = 0; i < R; i++) p.row_order[i] = (i & 1u) ? s0[R - 1 - i / 2] : s0[i / 2];          // heaviest, cheapest, ...
        else if (c->tune & 0x20000u) for (uint32_t i = 0; i < R; i++) p.row_order[i] = (uint16_t)((i & 1u) ? R / 2 + i / 2 : i / 2);   // top half / bottom half alternating
        else { uint32_t k = 0; for (uint32_t st = 0; st < 8; st++) for (uint32_t i = st; i < R; i += 8) p.row_order[k++] = s0[i]; } // stride 8 over the sorted order
    }
    RowOrderEntry &slot = c->row_cache[c->row_cache_next];
    c->row_cache_next = (c->row_cache_next + 1) % ROW_ORDER_CACHE;
    memcpy(slot.order, p.row_order, sizeof(p.row_order));
    memset(slot.key, 0, sizeof(slot.key));
    memcpy(slot.key, &key, sizeof(key));
    slot.valid = true;
}

// Make the workspace hold `items` half tiles and a pool of `pool` list entries (TileWorkspace, mil_internal.h).  Grows only;
// growing frees the old one (which waits for the kernels that use it) and allocates.
int workspace_reserve(mil_ctx *c, uint32_t items, uint32_t pool, bool in_frame_call)
{
    TileWorkspace &w = c->ws;
    if (items <= w.items_cap && pool <= w.pool_cap) return MIL_OK;
    if (w.pinned) return MIL_ERR_CAPACITY;          // a captured graph may hold the pointers (mil_light_ex_noalloc)
    items = items > w.items_cap ? items : w.items_cap;
    pool = pool > w.pool_cap ? pool : w.pool_cap;
    if (w.base) { HIP_TRY(c, hipFree(w.base)); w.base = nullptr; w.items_cap = 0; w.pool_cap = 0; }
    const size_t order_b = ((size_t)items * 4 + 255) & ~(size_t)255, meta_b = ((size_t)items * 8 + 255) & ~(size_t)255;
    const size_t pool_b = ((size_t)pool * 2 + 255) & ~(size_t)255;
    w.bytes = WS_COUNTER_BYTES + 2 * order_b + meta_b + pool_b;
    HIP_TRY(c, hipMalloc((void **)&w.base, w.bytes));
    w.hist = (uint32_t *)w.base;
    w.total = w.hist + MIL_WS_BUCKETS * MIL_WS_SHARDS;
    w.pool_used = w.total + 32;          // the next 128-byte line
    w.order = (uint32_t *)(w.base + WS_COUNTER_BYTES);
    w.rank = (uint32_t *)(w.base + WS_COUNTER_BYTES + order_b);
    w.meta = (uint2 *)(w.base + WS_COUNTER_BYTES + 2 * order_b);
    w.pool = (uint16_t *)(w.base + WS_COUNTER_BYTES + 2 * order_b + meta_b);
    w.items_cap = items; w.pool_cap = pool;
    w.list_cap = c->max_point < MIL_WS_LIST_MAX ? ((c->max_point + 63u) & ~63u) : MIL_WS_LIST_MAX;
    if (w.list_cap == 0) w.list_cap = 64u;
    w.used = false;
    if (in_frame_call) w.frame_call_allocs++;
    if (!w.done) HIP_TRY(c, hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    return MIL_OK;
}
uint32_t items_of(uint32_t width, uint32_t height)
{
    return 2u * ((width + MIL_TILE - 1) / MIL_TILE) * ((height + MIL_TILE - 1) / MIL_TILE);
}
uint32_t default_pool(uint32_t max_point)
{
    // measured: the 8K / 4096-light frame of BASELINE.json's cfg4 asks for 7.65 M entries (1870 per light), the 4K one for 2.2 M
    const uint64_t e = 3072ull * max_point;
    return (uint32_t)(e < (1u << 20) ? (1u << 20) : (e > (16u << 20) ? (16u << 20) : e));
}

// A launch was issued (status `launch`); remember on which stream the slot's content is being read.
int finish(mil_ctx *c, Slot *sl, hipStream_t s, hipError_t launch)
{
    if (launch != hipSuccess) return hip_fail(c, launch);
    if (!sl) return MIL_OK;
    int k = -1;
    for (int i = 0; i < SLOT_STREAMS && k < 0; i++)
        if (sl->done_used[i] && sl->done_stream[i] == s) k = i;
    for (int i = 0; i < SLOT_STREAMS && k < 0; i++)
        if (!sl->done_used[i]) k = i;
    if (k < 0) {        // four other streams are tracked: retire an entry whose kernel has finished, else wait for the oldest
        for (int i = 0; i < SLOT_STREAMS && k < 0; i++)
            if (hipEventQuery(sl->done[i]) == hipSuccess) k = i;
        if (k < 0) {
            HIP_TRY(c, hipEventSynchronize(sl->done[0]));
            k = 0;
        }
    }
    HIP_TRY(c, hipEventRecord(sl->done[k], s));
    sl->done_stream[k] = s;
    sl->done_used[k] = true;
    return MIL_OK;
}

} // namespace

extern "C" {

int mil_version(void) { return MIL_VERSION; }

const char *mil_strerror(int st)
{
    switch (st) {
    case MIL_OK: return "ok";
    case MIL_ERR_BAD_ARG: return "bad argument";
    case MIL_ERR_TOO_MANY_LIGHTS: return "more than MIL_MAX_DIR_LIGHTS directional lights";
    case MIL_ERR_HIP: return "HIP runtime error (see mil_last_hip_error)";
    case MIL_ERR_NO_DEVICE: return "no such HIP device";
    case MIL_ERR_CAPACITY: return "more point lights than the context capacity";
    case MIL_ERR_RCCL: return "RCCL error or librccl missing (see mil_mg_last_rccl_error)";
    default: return "unknown status";
    }
}

int mil_last_hip_error(const mil_ctx *c) { return c ? c->last_hip : 0; }

int mil_create(mil_ctx **out, int device, uint32_t max_point_lights)
{
    if (!out) return MIL_ERR_BAD_ARG;
    *out = nullptr;
    if (max_point_lights > 65535u) return MIL_ERR_BAD_ARG;   // tile lists hold 16-bit light indices
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MIL_ERR_NO_DEVICE;
    mil_ctx *c = new (std::nothrow) mil_ctx();
    if (!c) return MIL_ERR_HIP;
    c->device = device;
    c->max_point = max_point_lights;
    // Tuning knobs select kernels and paths the parity suite does not cover in every combination: a stray variable in the
    // host renderer's environment must not switch them, so they are read only under MIL_DEBUG=1.
    const char *dbg = getenv("MIL_DEBUG");
    if (dbg && atoi(dbg) == 1) {
        if (const char *e = getenv("MIL_TUNE")) c->tune = (uint32_t)atoi(e);
        if (const char *e = getenv("MIL_TONEMAP_ROWS")) c->tonemap_rows = (uint32_t)atoi(e);
        if (const char *e = getenv("MIL_FXAA_ROWS")) c->fxaa_rows = atoi(e) == 4 ? 4u : 2u;
        if (const char *e = getenv("MIL_K1_ROWS")) { int v = atoi(e); if (v >= 1 && v <= 64) c->k1_rows = (uint32_t)v; }
    }
    c->slot_bytes = 2 * (size_t)MIL_MAX_DIR_LIGHTS * 32 + (size_t)max_point_lights * (32 + 32 + 16);
    int rc = set_device(c);
    for (int i = 0; rc == MIL_OK && i < RING; i++) {
        Slot &s = c->ring[i];
        if (hipHostMalloc((void **)&s.host, c->slot_bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&s.dev, c->slot_bytes) != hipSuccess ||
            hipEventCreateWithFlags(&s.staged, hipEventDisableTiming) != hipSuccess)
            rc = MIL_ERR_HIP;
        for (int k = 0; rc == MIL_OK && k < SLOT_STREAMS; k++)
            if (hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming) != hipSuccess) rc = MIL_ERR_HIP;
    }
    if (rc == MIL_OK && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) rc = MIL_ERR_HIP;
    if (rc == MIL_OK && max_point_lights) rc = workspace_reserve(c, items_of(7680u, 4320u), default_pool(max_point_lights), false);
    if (rc == MIL_OK) rc = noalloc_init(c);      // the device address of the static scratch array of mil_light_ex_noalloc
    if (rc == MIL_OK) {
        uint8_t h[LUT11_N + LUT10_N];
        for (uint32_t i = 0; i < LUT11_N; i++) h[i] = srgb8(ufloat_dec(i, 6));
        for (uint32_t i = 0; i < LUT10_N; i++) h[LUT11_N + i] = srgb8(ufloat_dec(i, 5));
        if (hipMalloc((void **)&c->lut, sizeof(h)) != hipSuccess ||
            hipMemcpy(c->lut, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess)
            rc = MIL_ERR_HIP;
    }
    if (rc != MIL_OK) {
        mil_destroy(c);
        return rc;
    }
    *out = c;
    return MIL_OK;
}

void mil_destroy(mil_ctx *c)
{
    if (!c) return;
    set_device(c);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (Slot &s : c->ring) {
        (void)slot_wait_readers(c, s);
        for (int k = 0; k < SLOT_STREAMS; k++)
            if (s.done[k]) (void)hipEventDestroy(s.done[k]);
        if (s.staged) (void)hipEventDestroy(s.staged);
        if (s.host) (void)hipHostFree(s.host);
        if (s.dev) (void)hipFree(s.dev);
    }
    if (c->lut) (void)hipFree(c->lut);
    if (c->ws.base) (void)hipFree(c->ws.base);
    if (c->ws.done) (void)hipEventDestroy(c->ws.done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

int mil_set_tile_count_buffer(mil_ctx *c, uint32_t *counts, uint32_t n_tiles)
{
    if (!c) return MIL_ERR_BAD_ARG;
    c->tile_counts = counts;
    c->tile_counts_cap = counts ? n_tiles : 0;
    return MIL_OK;
}

int mil_set_two_kernel_threshold(mil_ctx *c, uint32_t min_point_lights)
{
    if (!c) return MIL_ERR_BAD_ARG;
    c->two_kernel_min = min_point_lights;
    return MIL_OK;
}

int mil_set_tile_stamp_buffer(mil_ctx *c, uint32_t *stamps, uint32_t n_tiles)
{
    if (!c) return MIL_ERR_BAD_ARG;
    c->tile_stamps = stamps;
    c->tile_stamps_cap = stamps ? n_tiles : 0;
    return MIL_OK;
}

int mil_row_weights(const mil_frame_matrix_data *frame, const mil_push_constants *push, uint32_t gb_height, uint32_t y_offset,
                    const mil_band_layout *bands, const mil_point_light *pl, uint32_t n_point, uint32_t *weights)
{
    if (!frame || !push || !weights || gb_height == 0 || (n_point && !pl)) return MIL_ERR_BAD_ARG;
    if (bands && (!bands_ok(bands) || y_offset != 0)) return MIL_ERR_BAD_ARG;
    if (!(push->resolution[1] >= 1.0f)) return MIL_ERR_BAD_ARG;
    row_weights(frame, push->resolution[1], gb_height, y_offset, bands, pl, n_point, weights);
    return MIL_OK;
}

uint32_t mil_band_local_rows(const mil_band_layout *b, uint32_t height)
{
    if (!b || !bands_ok(b)) return 0;
    const uint64_t nb = ((uint64_t)height + b->band_rows - 1) / b->band_rows;        // bands of the frame, the last may be short
    if (b->rank >= nb) return 0;
    const uint64_t mine = (nb - b->rank + b->n_ranks - 1) / b->n_ranks;              // bands rank, rank + n, ...
    uint64_t rows = mine * b->band_rows;
    if ((nb - 1) % b->n_ranks == b->rank) rows -= nb * b->band_rows - height;       // the short last band is this rank's
    return (uint32_t)rows;
}

int mil_reserve(mil_ctx *c, uint32_t width, uint32_t height)
{
    if (!c || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 30)) return MIL_ERR_BAD_ARG;
    if (c->max_point == 0) return MIL_OK;               // no point lights, no tile lists
    int rc = set_device(c);
    if (rc) return rc;
    return workspace_reserve(c, items_of(width, height), c->ws.pool_cap ? c->ws.pool_cap : default_pool(c->max_point), false);
}

int mil_set_list_pool(mil_ctx *c, uint32_t entries)
{
    if (!c || entries == 0 || entries > (1u << 30)) return MIL_ERR_BAD_ARG;
    if (c->max_point == 0) return MIL_OK;
    int rc = set_device(c);
    if (rc) return rc;
    TileWorkspace &w = c->ws;
    if (entries == w.pool_cap) return MIL_OK;
    if (w.pinned) return MIL_ERR_CAPACITY;
    const uint32_t items = w.items_cap ? w.items_cap : items_of(7680u, 4320u);
    // a smaller pool is a new allocation too: drop the old one first (hipFree waits for its kernels)
    if (w.base) { HIP_TRY(c, hipFree(w.base)); w.base = nullptr; w.items_cap = 0; w.pool_cap = 0; }
    return workspace_reserve(c, items, entries, false);
}

int mil_workspace_info(mil_ctx *c, mil_workspace_stats *out)
{
    if (!c || !out) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    const TileWorkspace &w = c->ws;
    memset(out, 0, sizeof(*out));
    out->bytes = (uint64_t)w.bytes;
    out->half_tiles = w.items_cap; out->pool_entries = w.pool_cap; out->list_entries_max = w.list_cap;
    out->frame_call_allocations = w.frame_call_allocs;
    if (w.base && w.used) {
        HIP_TRY(c, hipEventSynchronize(w.done));
        uint32_t v[32 * (MIL_WS_POOL_PARTS + 2)];
        HIP_TRY(c, hipMemcpy(v, w.total, sizeof(v), hipMemcpyDeviceToHost));
        out->last_items = v[0];
        for (uint32_t k = 0; k < MIL_WS_POOL_PARTS; k++) out->last_pool_entries += v[32 * (1 + k)];
        out->last_brute_macro_tiles = v[32 * (1 + MIL_WS_POOL_PARTS)];
    }
    return MIL_OK;
}

int mil_sync(mil_ctx *c, void *stream)
{
    if (!c) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));
    return MIL_OK;
}

} // extern "C"

// mil_light_ex, mil_shade_frame, mil_shade_bands: the same launch, the later ones with the later subpasses folded into the
// store / the rows dealt in bands.  Every argument is validated before anything is staged or enqueued: an argument error
// has no side effect.
int milhost::shade_check(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                         const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *ex,
                         const mil_frame_passes *passes, void *out, uint32_t out_pitch_px, mil_out_format fmt,
                         const mil_band_layout *bands, bool out_frame, const PreStaged *pre)
{
    if (!c || !frame || !out) return MIL_ERR_BAD_ARG;
    if (passes && ((passes->accum != nullptr) != (passes->reveal != nullptr))) return MIL_ERR_BAD_ARG;
    int rc = check_gbuffer(gb, push, bands);
    if (rc) return rc;
    if (out_pitch_px < gb->width || (unsigned)fmt > MIL_OUT_SRGB8_BGRA_VIA_LDR) return MIL_ERR_BAD_ARG;
    if (push->light_count > MIL_MAX_DIR_LIGHTS) return MIL_ERR_TOO_MANY_LIGHTS;
    if (push->light_count && !lights) return MIL_ERR_BAD_ARG;
    const uint32_t n_point = ex ? ex->n_point : 0;
    if (n_point > (pre ? pre->capacity : c->max_point)) return MIL_ERR_CAPACITY;
    if (n_point && !ex->point_lights) return MIL_ERR_BAD_ARG;
    const mil_shadow *sh = ex ? ex->shadow : nullptr;
    if (sh && (!sh->map || sh->size == 0 || sh->size > 16384u)) return MIL_ERR_BAD_ARG;   // 32-bit texel offsets in the taps
    if (passes && passes->accum && (passes->accum_pitch_px < gb->width || passes->reveal_pitch_px < gb->width)) return MIL_ERR_BAD_ARG;
    if (out_frame && !bands) return MIL_ERR_BAD_ARG;
    if (ex && ex->ldr_out && (fmt != MIL_OUT_HDR_RGBA32F || ex->ldr_out_pitch_px < gb->width)) return MIL_ERR_BAD_ARG;
    return MIL_OK;
}

int milhost::shade(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                   const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *ex,
                   const mil_frame_passes *passes, void *out, uint32_t out_pitch_px, mil_out_format fmt, uint32_t flags, void *stream,
                   const mil_band_layout *bands, bool out_frame, const PreStaged *pre)
{
    int rc = shade_check(c, gb, frame, lights, push, ex, passes, out, out_pitch_px, fmt, bands, out_frame, pre);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    const uint32_t n_point = ex ? ex->n_point : 0;
    const mil_shadow *sh = ex ? ex->shadow : nullptr;

    hipStream_t s = (hipStream_t)stream;
    const bool strict = (flags & MIL_FLAG_STRICT) != 0;
    mil::ShadeParams p;
    fill_common(p, gb, frame, push, bands);
    p.rows_per_block = c->k1_rows;
    p.tune = c->tune;
    Slot *slot = nullptr;
    if (pre) {          // the light arrays are on the device already (prepared by a kernel earlier on this stream)
        inline_dir_records(p, lights, push->light_count, strict);
        p.n_dir = push->light_count; p.n_point = n_point;
        slot_pointers(p, pre->dev, pre->capacity);
    } else if ((rc = stage_lights(c, lights, push->light_count, ex ? ex->point_lights : nullptr, n_point, strict, s, p, slot))) {
        return rc;
    }
    if (ex && (ex->ambient[0] != 0.0f || ex->ambient[1] != 0.0f || ex->ambient[2] != 0.0f)) {
        p.has_ambient = 1;
        memcpy(p.ambient, ex->ambient, 12);
    }
    if (sh) {
        p.has_shadow = 1;
        p.shadow_size = sh->size;
        p.shadow_bias = sh->bias;
        p.shadow_map = sh->map;
        p.shadow_lds = (flags & MIL_FLAG_SHADOW_LDS) ? 1u : 0u;
        memcpy(p.lvp, sh->light_view_proj, 64);
    }
    p.out = out;
    p.out_pitch = out_pitch_px;
    if (out_frame) {    // rows of the assembled frame: the dealing's row map applies to the store too
        p.out_band_extra = p.band_extra;
        p.out = (uint8_t *)out + (size_t)p.y_offset * out_pitch_px * (fmt == MIL_OUT_HDR_RGBA32F ? 16u : 4u);
    }
    if (ex && ex->ldr_out) {
        p.out2 = ex->ldr_out + (out_frame ? (size_t)p.y_offset * ex->ldr_out_pitch_px : 0);
        p.out2_pitch = ex->ldr_out_pitch_px;
    }
    p.lut11 = c->lut;
    p.lut10 = c->lut + LUT11_N;
    const bool whole = passes != nullptr;
    if (whole) {
        if (passes->sky) {
            p.has_sky = 1;
            static_assert(sizeof(p.hw) == sizeof(mil_hosek_wilkie_params), "HosekWilkieParams is 11 x vec4");
            memcpy(p.hw, passes->sky, sizeof(p.hw));
        }
        p.accum = (const uint16_t *)passes->accum; p.reveal = (const uint16_t *)passes->reveal;
        p.accum_pitch = passes->accum_pitch_px; p.reveal_pitch = passes->reveal_pitch_px;
        p.resolve_dma = p.accum && !(gb->width & 1u) && !(p.reveal_pitch & 1u) && !((uintptr_t)p.reveal & 3u) && !((uintptr_t)p.accum & 3u) && !(c->tune & 128u);
    }

    const bool tiled = (n_point > 0 && !(flags & MIL_FLAG_NO_CULL)) || (p.has_shadow && p.shadow_lds && p.n_dir > 0);
    hipError_t le;
    const uint32_t n_tiles = p.tiles_x * p.tiles_y;            // < 2^22: pitch * height < 2^30
    if (tiled) {
        p.tile_counts = n_tiles <= c->tile_counts_cap ? c->tile_counts : nullptr;
        p.tile_stamps = n_tiles <= c->tile_stamps_cap ? c->tile_stamps : nullptr;
    }
    // The two-kernel form (k_bin -> k_shade) serves the fast-mode launches with many point lights (mil_light_ex, mil_shade_bands
    // and mil_shade_frame alike); strict mode, LDS-staged shadows and the wave-stamp diagnostics stay on k_tiled (MIL_TUNE bit
    // 1024 forces it).
    // (MIL_TUNE bit 2048: the stamp buffer, then 8 u32 per tile, goes to k_shade instead: one record per half tile)
    if (tiled && (c->tune & 2048u)) p.tile_stamps = 2u * n_tiles <= c->tile_stamps_cap ? c->tile_stamps : nullptr;
    // Below ~900 point lights k_tiled is the faster of the two on the 4K frame (256 lights: 101 us against 122; 1024: 234
    // against 226; 4096: 625 against 487): the pre-pass costs what culling and sky tiles cost there.  The threshold is a
    // context setting (mil_set_two_kernel_threshold; the parity tests set 1 so that small frames cover both kernels).
    // A whole-frame launch with a sky or a resolve subpass makes every half tile an item (k_shade's store does those passes),
    // which costs the two-kernel form what the sky tiles cost: its break-even is at twice the light count there (4K, measured:
    // 1536 lights 314 us against 299 for k_tiled, 2048 lights 366 against 378, 4096 lights 526 against 624).
    const bool frame_items = whole && (p.has_sky || p.accum);
    const uint64_t two_min = frame_items ? 2ull * c->two_kernel_min : c->two_kernel_min;
    const bool many = n_point >= two_min;
    bool two = tiled && many && !strict && !(flags & MIL_FLAG_NO_CULL) && !p.shadow_lds && (!p.tile_stamps || (c->tune & 2048u)) && !(c->tune & 1024u);
    if (two) {
        const uint32_t items = 2u * n_tiles;
        if (pre || c->ws.pinned)      // the entry point that allocates nothing never grows the workspace; a pinned one is never freed
            two = items <= c->ws.items_cap;
        else if ((rc = workspace_reserve(c, items, c->ws.pool_cap ? c->ws.pool_cap : default_pool(c->max_point), true))) return rc;
    }
    if (two) {
        TileWorkspace &w = c->ws;
        // One workspace per context: a launch on another stream than the last one waits for it.  Under stream capture
        // (mil_light_ex_noalloc inside hipStreamBeginCapture) events of the outside world may not be waited for or recorded:
        // the captured launch stays outside this protocol - replaying the graph beside other shading calls of the same
        // context is the caller's to order - and the workspace is pinned, because the graph holds its pointers.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (pre && s && hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
        const bool capturing = cap != hipStreamCaptureStatusNone;
        if (capturing) w.pinned = true;
        if (!capturing && w.used && w.last_stream != s) HIP_TRY(c, hipStreamWaitEvent(s, w.done, 0));
        p.ws_hist = w.hist; p.ws_rank = w.rank; p.ws_total = w.total; p.ws_pool_used = w.pool_used; p.ws_order = w.order;
        p.ws_meta = w.meta; p.ws_pool = w.pool;
        p.ws_items_cap = w.items_cap; p.ws_pool_cap = w.pool_cap;
        const uint32_t want = (n_point + 63u) & ~63u;
        p.ws_list_cap = want < w.list_cap ? want : w.list_cap;
        HIP_TRY(c, hipMemsetAsync(w.hist, 0, WS_COUNTER_BYTES, s));
        le = mil::fast::launch_bin((int)fmt, p, s);
        if (le == hipSuccess) le = mil::fast::launch_order(p, s);
        if (le == hipSuccess) le = mil::fast::launch_shade((int)fmt, p, whole, s);
        if (le == hipSuccess && !capturing) {
            HIP_TRY(c, hipEventRecord(w.done, s));
            w.used = true; w.last_stream = s;
        }
    } else if (tiled) {
        if (!pre) order_rows(c, slot, p, frame, bands, gb->y_offset, ex ? ex->point_lights : nullptr, n_point);
        le = strict ? mil::strict::launch_tiled((int)fmt, p, whole, s) : mil::fast::launch_tiled((int)fmt, p, whole, s);
    } else {
        le = strict ? mil::strict::launch_stream((int)fmt, p, whole, s) : mil::fast::launch_stream((int)fmt, p, whole, s);
    }
    return finish(c, slot, s, le);
}

extern "C" {

int mil_light_ex(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                 const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *ex,
                 void *out, uint32_t out_pitch_px, mil_out_format fmt, uint32_t flags, void *stream)
{
    return shade(c, gb, frame, lights, push, ex, nullptr, out, out_pitch_px, fmt, flags, stream, nullptr, false, nullptr);
}

static const mil_frame_passes no_passes = {};

int mil_shade_frame(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                    const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *ex,
                    const mil_frame_passes *passes, void *out, uint32_t out_pitch_px, mil_out_format fmt,
                    uint32_t flags, void *stream)
{
    return shade(c, gb, frame, lights, push, ex, passes ? passes : &no_passes, out, out_pitch_px, fmt, flags, stream, nullptr, false, nullptr);
}

int mil_shade_bands(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                    const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *ex,
                    const mil_frame_passes *passes, const mil_band_layout *bands, void *out, uint32_t out_pitch_px,
                    int out_is_frame, mil_out_format fmt, uint32_t flags, void *stream)
{
    // passes == NULL is the lighting launch alone (as mil_light_ex), not a whole frame without later subpasses: the same
    // pixels either way, but only the former may take the two-kernel form
    return shade(c, gb, frame, lights, push, ex, passes, out, out_pitch_px, fmt, flags, stream, bands,
                 bands != nullptr && out_is_frame != 0, nullptr);
}

int mil_light_directional(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                          const mil_directional_light *lights, const mil_push_constants *push, float *hdr,
                          uint32_t hdr_pitch_px, uint32_t flags, void *stream)
{
    return mil_light_ex(c, gb, frame, lights, push, nullptr, hdr, hdr_pitch_px, MIL_OUT_HDR_RGBA32F,
                        flags & MIL_FLAG_STRICT, stream);
}

int mil_tonemap(mil_ctx *c, const float *hdr, uint32_t hdr_pitch_px, uint32_t width, uint32_t height, uint32_t *out,
                uint32_t out_pitch_px, mil_out_format fmt, uint32_t flags, void *stream)
{
    if (!c || !hdr || !out || width == 0 || height == 0) return MIL_ERR_BAD_ARG;
    if (hdr_pitch_px < width || out_pitch_px < width) return MIL_ERR_BAD_ARG;
    if (fmt != MIL_OUT_LDR_R11G11B10F && fmt != MIL_OUT_SRGB8_BGRA && fmt != MIL_OUT_SRGB8_BGRA_VIA_LDR) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    mil::TonemapParams p;
    p.hdr = (const float4 *)hdr; p.out = out; p.width = width; p.height = height;
    p.hdr_pitch = hdr_pitch_px; p.out_pitch = out_pitch_px;
    p.lut11 = c->lut; p.lut10 = c->lut + LUT11_N;
    p.rows = c->tonemap_rows;
    const hipError_t le = (flags & MIL_FLAG_STRICT) ? mil::strict::launch_tonemap((int)fmt, p, (hipStream_t)stream) : mil::fast::launch_tonemap((int)fmt, p, (hipStream_t)stream);
    return finish(c, nullptr, (hipStream_t)stream, le);
}

int mil_fxaa(mil_ctx *c, const uint32_t *ldr, uint32_t ldr_pitch_px, uint32_t width, uint32_t height, uint32_t row_begin,
             uint32_t row_count, uint32_t *out, uint32_t out_pitch_px, uint32_t flags, void *stream)
{
    if (!c || !ldr || !out || width == 0 || height == 0) return MIL_ERR_BAD_ARG;
    if (ldr_pitch_px < width || out_pitch_px < width) return MIL_ERR_BAD_ARG;
    if (row_count == 0 || (uint64_t)row_begin + row_count > height) return MIL_ERR_BAD_ARG;
    if ((uint64_t)ldr_pitch_px * height >= (1ull << 30) || height >= (1u << 24) || ldr_pitch_px >= (1u << 24)) return MIL_ERR_BAD_ARG;   // 32-bit texel offsets in the kernels
    int rc = set_device(c);
    if (rc) return rc;
    mil::FxaaParams p;
    p.ldr = ldr; p.out = out; p.width = width; p.height = height; p.ldr_pitch = ldr_pitch_px; p.out_pitch = out_pitch_px;
    p.row_begin = row_begin; p.row_count = row_count;
    p.lut11 = c->lut; p.lut10 = c->lut + LUT11_N;
    p.rows = c->fxaa_rows;
    const hipError_t le = (flags & MIL_FLAG_STRICT) ? mil::strict::launch_fxaa(p, (hipStream_t)stream) : mil::fast::launch_fxaa(p, (hipStream_t)stream);
    return finish(c, nullptr, (hipStream_t)stream, le);
}

int mil_sky_hosek(mil_ctx *c, const float *depth, uint32_t depth_pitch_px, uint32_t width, uint32_t height, uint32_t y_offset,
                  const mil_frame_matrix_data *frame, const mil_push_constants *push, const mil_hosek_wilkie_params *params,
                  float *hdr, uint32_t hdr_pitch_px, uint32_t flags, void *stream)
{
    if (!c || !depth || !frame || !push || !params || !hdr || width == 0 || height == 0) return MIL_ERR_BAD_ARG;
    if (depth_pitch_px < width || hdr_pitch_px < width) return MIL_ERR_BAD_ARG;
    if (!(push->resolution[0] >= 1.0f) || !(push->resolution[1] >= 1.0f) || (float)width > push->resolution[0]) return MIL_ERR_BAD_ARG;
    if ((double)y_offset + (double)height > (double)push->resolution[1]) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    mil::SkyParams p;
    memset(&p, 0, sizeof(p));
    p.depth = depth; p.hdr = hdr; p.width = width; p.height = height; p.depth_pitch = depth_pitch_px; p.hdr_pitch = hdr_pitch_px;
    p.y_offset = y_offset;
    memcpy(p.inv_proj, frame->inv_projection, 64); memcpy(p.inv_view, frame->inv_view, 64); memcpy(p.cam, frame->camera_position, 12);
    p.res_x = push->resolution[0]; p.res_y = push->resolution[1];
    fold_position_matrix(frame, p.res_x, p.res_y, p.pos_m);
    memcpy(p.hw, params, sizeof(p.hw));
    const hipError_t le = (flags & MIL_FLAG_STRICT) ? mil::strict::launch_sky(p, (hipStream_t)stream) : mil::fast::launch_sky(p, (hipStream_t)stream);
    return finish(c, nullptr, (hipStream_t)stream, le);
}

int mil_transparency_resolve(mil_ctx *c, const void *accum, uint32_t accum_pitch_px, const void *reveal, uint32_t reveal_pitch_px,
                             uint32_t width, uint32_t height, float *hdr, uint32_t hdr_pitch_px, uint32_t flags, void *stream)
{
    if (!c || !accum || !reveal || !hdr || width == 0 || height == 0) return MIL_ERR_BAD_ARG;
    if (accum_pitch_px < width || reveal_pitch_px < width || hdr_pitch_px < width) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    mil::ResolveParams p;
    p.accum = (const uint16_t *)accum; p.reveal = (const uint16_t *)reveal; p.hdr = hdr; p.width = width; p.height = height;
    p.accum_pitch = accum_pitch_px; p.reveal_pitch = reveal_pitch_px; p.hdr_pitch = hdr_pitch_px;
    const hipError_t le = (flags & MIL_FLAG_STRICT) ? mil::strict::launch_resolve(p, (hipStream_t)stream) : mil::fast::launch_resolve(p, (hipStream_t)stream);
    return finish(c, nullptr, (hipStream_t)stream, le);
}

int mil_transparency_accumulate(mil_ctx *c, const mil_transparent_fragments *fr, const mil_frame_matrix_data *frame,
                                const mil_directional_light *lights, const mil_push_constants *push, float *accum,
                                float *reveal, uint32_t flags, void *stream)
{
    if (!c || !fr || !frame || !push) return MIL_ERR_BAD_ARG;
    if (push->light_count > MIL_MAX_DIR_LIGHTS) return MIL_ERR_TOO_MANY_LIGHTS;
    if (fr->count == 0) return MIL_OK;                                   // no translucent fragment this frame
    if (fr->count >= (1u << 30)) return MIL_ERR_BAD_ARG;
    if (!fr->position_depth || !fr->normal_opacity || !fr->albedo_roughness || !fr->metallic || !accum || !reveal) return MIL_ERR_BAD_ARG;
    if (push->light_count && !lights) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    mil::AccumParams p;
    memset(&p, 0, sizeof(p));
    p.pos_z = (const float4 *)fr->position_depth; p.nrm_op = (const float4 *)fr->normal_opacity;
    p.alb_rough = (const float4 *)fr->albedo_roughness; p.metallic = fr->metallic;
    p.accum = (float4 *)accum; p.reveal = reveal; p.count = fr->count; p.n_dir = push->light_count;
    memcpy(p.cam, frame->camera_position, 12);
    for (uint32_t i = 0; i < p.n_dir && i < MIL_DIR_INLINE; i++) memcpy(p.dir_inline[i], &lights[i], 28);
    // Beyond the inline records the raw array is read from a staged slot: the one the lighting call of this frame staged
    // when it holds the same directional lights (the reference binds one lights UBO to both draws, render/mod.rs:252)
    Slot *slot = nullptr;
    if (p.n_dir > MIL_DIR_INLINE) {
        const int base = c->cur < 0 ? 0 : c->cur;            // the frame's lighting call staged them most likely
        for (int k = 0; !slot && k < RING; k++) {
            Slot &cs = c->ring[(base + k) % RING];
            if (!cs.valid || cs.n_dir != p.n_dir) continue;
            bool same = true;
            for (uint32_t i = 0; same && i < p.n_dir; i++) same = !memcmp(cs.host + 32 * i, &lights[i], 28);
            if (same) {
                HIP_TRY(c, hipStreamWaitEvent(s, cs.staged, 0));
                slot = &cs;
                p.dir_raw = (const float4 *)cs.dev;
            }
        }
        if (!slot) {
            mil::ShadeParams sp;
            if ((rc = stage_lights(c, lights, p.n_dir, nullptr, 0, true, s, sp, slot))) return rc;
            p.dir_raw = sp.dir_raw;
        }
    }
    const hipError_t le = (flags & MIL_FLAG_STRICT) ? mil::strict::launch_accumulate(p, s) : mil::fast::launch_accumulate(p, s);
    return finish(c, slot, s, le);
}

int mil_import_external_memory(mil_ctx *c, int fd, uint64_t size, mil_external_memory **mem, void **dev_ptr)
{
    if (!c || fd < 0 || size == 0 || !mem || !dev_ptr) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    hipExternalMemoryHandleDesc d;
    memset(&d, 0, sizeof(d));
    d.type = hipExternalMemoryHandleTypeOpaqueFd;
    d.handle.fd = fd;
    d.size = size;
    mil_external_memory *m = new (std::nothrow) mil_external_memory();
    if (!m) return MIL_ERR_HIP;
    m->imported = true; m->size = (size_t)size;
    hipError_t e = hipImportExternalMemory(&m->ext, &d);
    if (e == hipSuccess) {
        hipExternalMemoryBufferDesc b;
        memset(&b, 0, sizeof(b));
        b.size = size;
        e = hipExternalMemoryGetMappedBuffer(&m->mapped, m->ext, &b);
        if (e != hipSuccess) (void)hipDestroyExternalMemory(m->ext);
    }
    if (e != hipSuccess) { delete m; return hip_fail(c, e); }
    *mem = m; *dev_ptr = m->mapped;
    return MIL_OK;
}

int mil_alloc_exportable_memory(mil_ctx *c, uint64_t size, mil_external_memory **mem, void **dev_ptr, int *fd, uint64_t *allocated_size)
{
    if (!c || size == 0 || !mem || !dev_ptr || !fd) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    HIP_TRY(c, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran == 0) gran = 4096;
    mil_external_memory *m = new (std::nothrow) mil_external_memory();
    if (!m) return MIL_ERR_HIP;
    m->size = ((size_t)size + gran - 1) / gran * gran;
    hipError_t e = hipMemCreate(&m->handle, m->size, &prop, 0);
    m->have_handle = e == hipSuccess;
    if (e == hipSuccess) e = hipMemAddressReserve(&m->va, m->size, 0, nullptr, 0);
    if (e == hipSuccess) { e = hipMemMap(m->va, m->size, 0, m->handle, 0); m->is_mapped = e == hipSuccess; }
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof(acc));
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(m->va, m->size, &acc, 1);
    }
    int out_fd = -1;
    if (e == hipSuccess) e = hipMemExportToShareableHandle(&out_fd, m->handle, hipMemHandleTypePosixFileDescriptor, 0);
    if (e != hipSuccess) { (void)mil_release_external_memory(c, m); return hip_fail(c, e); }
    *mem = m; *dev_ptr = m->va; *fd = out_fd;
    if (allocated_size) *allocated_size = m->size;
    return MIL_OK;
}

int mil_release_external_memory(mil_ctx *c, mil_external_memory *m)
{
    if (!c || !m) return MIL_ERR_BAD_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    hipError_t first = hipSuccess;
    auto keep = [&](hipError_t e) { if (first == hipSuccess) first = e; };
    if (m->imported) {
        if (m->mapped) keep(hipFree(m->mapped));
        if (m->ext) keep(hipDestroyExternalMemory(m->ext));
    } else {
        if (m->is_mapped) keep(hipMemUnmap(m->va, m->size));
        if (m->va) keep(hipMemAddressFree(m->va, m->size));
        if (m->have_handle) keep(hipMemRelease(m->handle));
    }
    delete m;
    return first == hipSuccess ? MIL_OK : hip_fail(c, first);
}

int mil_count_pairs(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame, const mil_push_constants *push,
                    const mil_point_light *pl, uint32_t n_point, unsigned long long *count, void *stream)
{
    if (!c || !frame || !count) return MIL_ERR_BAD_ARG;
    int rc = check_gbuffer(gb, push, nullptr);
    if (rc) return rc;
    if (n_point > c->max_point) return MIL_ERR_CAPACITY;
    if (n_point && !pl) return MIL_ERR_BAD_ARG;
    if ((rc = set_device(c))) return rc;
    hipStream_t s = (hipStream_t)stream;
    mil::ShadeParams p;
    fill_common(p, gb, frame, push, nullptr);
    Slot *slot = nullptr;
    if ((rc = stage_lights(c, nullptr, 0, pl, n_point, false, s, p, slot))) return rc;
    HIP_TRY(c, hipMemsetAsync(count, 0, sizeof(unsigned long long), s));
    return finish(c, slot, s, mil::fast::launch_count_pairs(p, count, s));
}

} // extern "C"
