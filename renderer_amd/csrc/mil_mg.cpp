// Multi-GPU layer of libmil.so (include/mil.h, mil_mg_*): one frame cut over the GPUs of a node (strips: contiguous or
// dealt bands) or one frame per GPU, every rank's rows gathered to GPU 0.
//
// The reference renders on ONE device (render/vulkan.rs:106-108 picks a PhysicalDevice, render/mod.rs:189-219 records
// the lighting draw on its only queue); there is nothing to be at parity with except that the assembled frame equals the
// single-GPU frame bit for bit.  The pass has no data-path exchange (every pixel reads its own texel + replicated
// uniforms); the only communication is the gather, and RCCL has no gather primitive, so it is grouped ncclSend / ncclRecv:
// every sender uses its own xGMI link to GPU 0.  A rank's rows are shaded in `chunks` launches on its compute stream;
// the gather of chunk k runs on a second stream behind an event, beside the shading of chunk k + 1.
//
// RCCL is bound at run time (dlopen "librccl.so.1": the instance the host process already loaded, e.g. torch's, else the
// system one), so single-GPU users of libmil.so need no RCCL; a missing library is MIL_ERR_RCCL, never a fallback.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <new>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "mil_internal.h"

using namespace milhost;

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.so) break;
    }
    if (!r.so) return r;
#define MIL_SYM(field, sym) *(void **)&r.field = dlsym(r.so, sym); if (!r.field) return r;
    MIL_SYM(GetUniqueId, "ncclGetUniqueId") MIL_SYM(CommInitAll, "ncclCommInitAll") MIL_SYM(CommInitRank, "ncclCommInitRank")
    MIL_SYM(CommDestroy, "ncclCommDestroy") MIL_SYM(CommCount, "ncclCommCount") MIL_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    MIL_SYM(GroupStart, "ncclGroupStart") MIL_SYM(GroupEnd, "ncclGroupEnd") MIL_SYM(Send, "ncclSend") MIL_SYM(Recv, "ncclRecv")
#undef MIL_SYM
    r.ok = true;
    return r;
}

static_assert(sizeof(ncclUniqueId) == MIL_MG_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

// A run of rows that is contiguous on both sides of the gather: local rows [local0, local0 + rows) of a rank are frame
// rows [frame0, frame0 + rows).
struct Piece { uint32_t local0, rows, frame0; };

struct Local {                      // one rank this process drives
    int rank = 0, device = 0;
    // Two shading lanes (context + stream), taken in turn frame by frame: a launch spends its last ~12 us below half
    // occupancy (the heavy tiles' tail) and its first few filling the chip, so with consecutive frames on alternating streams
    // the ramp of one runs under the tail of the other (one GPU, 4K: 256 lights 86 -> 72 us per frame, 4096 lights 424 -> 390).
    // A context per lane because a context's workspace and light staging serve one launch at a time.  MIL_MG_LANES=1 in
    // the environment keeps every frame on the first lane.
    static constexpr int LANES = 2;
    mil_ctx *ctx[LANES] = {};
    hipStream_t compute[LANES] = {}, gather = nullptr;
    int lanes = LANES;
    bool lanes_overlap = false;     // the probe at creation saw the two lane streams run side by side
    uint64_t frames = 0;
    ncclComm_t comm = nullptr;
    std::vector<hipEvent_t> chunk_done;      // compute -> gather, one per chunk
    // The last FRAMES_IN_FLIGHT frames of this rank: an event on the gather stream behind the frame's last transfer, and the
    // buffers the frame's shading wrote / its transfers read or wrote.  A new frame waits only for earlier frames that touch
    // one of ITS buffers: a caller that rotates its output buffers gets the gather of frame i under the shading of frame
    // i + 1; a caller that reuses one buffer gets them one after the other.
    static constexpr int FRAMES_IN_FLIGHT = 4;
    hipEvent_t gathered[FRAMES_IN_FLIGHT] = {};
    std::vector<const void *> touched[FRAMES_IN_FLIGHT];
    bool gathered_valid[FRAMES_IN_FLIGHT] = {};
    int frame_next = 0;
};

} // namespace

struct mil_mg {
    int world = 0;
    mil_mg_config cfg = {};
    std::vector<Local> locals;
    int rccl_error = 0, last_hip = 0, rccl_ranks = 0;
    bool single_process = false;
    // rank 0's device: where the dealt rows of the other ranks land before they are scattered into the frame (one slice per rank)
    uint8_t *stage = nullptr;
    size_t stage_slice = 0;
    // Recorded on rank 0's gather stream behind the scatters of a frame: the next frame's copies into the staging slices (which
    // run on the SENDERS' gather streams with the peer-copy transport) wait for it, so a slice is never overwritten before it
    // was scattered.  (The RCCL receives sit on rank 0's gather stream themselves and are ordered behind the scatters anyway.)
    hipEvent_t stage_free = nullptr;
    bool stage_free_valid = false;
};

namespace {

int hipf(mil_mg *m, hipError_t e) { m->last_hip = (int)e; return MIL_ERR_HIP; }
int ncclf(mil_mg *m, ncclResult_t e) { m->rccl_error = (int)e; return MIL_ERR_RCCL; }
#define MG_HIP(m, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hipf((m), e_); } while (0)
#define MG_NCCL(m, expr) do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) return ncclf((m), e_); } while (0)

bool cfg_ok(const mil_mg_config *c)
{
    if (c->band_rows && (c->band_rows < MIL_TILE || (c->band_rows & (c->band_rows - 1u)) || c->band_rows > 4096u)) return false;
    return c->chunks <= 64 && c->transport <= MIL_MG_TRANSPORT_PEER_COPY && c->max_point_lights <= 65535u;
}

// The rows of `rank`: contiguous strip on a 16-row cut (band_rows = 0; 2160 -> 7 x 272 + 256, SURVEY.md 8 e) or every
// world-th band.  Local rows are the pieces back to back.
void rank_pieces(const mil_mg_config &cfg, int world, int rank, uint32_t height, std::vector<Piece> &out)
{
    out.clear();
    if (cfg.band_rows == 0) {
        const uint64_t per = (((uint64_t)height + world - 1) / world + MIL_TILE - 1) / MIL_TILE * MIL_TILE;
        const uint64_t y0 = (uint64_t)rank * per < height ? (uint64_t)rank * per : height;
        const uint64_t y1 = (uint64_t)(rank + 1) * per < height ? (uint64_t)(rank + 1) * per : height;
        if (y1 > y0) out.push_back(Piece{ 0u, (uint32_t)(y1 - y0), (uint32_t)y0 });
        return;
    }
    const uint32_t B = cfg.band_rows, nb = (height + B - 1) / B;
    uint32_t local = 0;
    for (uint32_t b = (uint32_t)rank; b < nb; b += (uint32_t)world) {
        const uint32_t rows = (b + 1) * B <= height ? B : height - b * B;
        out.push_back(Piece{ local, rows, b * B });
        local += rows;
    }
}
uint32_t rank_rows(const mil_mg_config &cfg, int world, int rank, uint32_t height)
{
    std::vector<Piece> p;
    rank_pieces(cfg, world, rank, height, p);
    return p.empty() ? 0u : p.back().local0 + p.back().rows;
}
// Local row ranges of the `chunks` launches of a rank: cuts on multiples of the band (or of 16 rows), as even as that allows.
void chunk_cuts(const mil_mg_config &cfg, uint32_t rows, std::vector<uint32_t> &cuts)
{
    const uint32_t n = cfg.chunks ? cfg.chunks : 4u, g = cfg.band_rows ? cfg.band_rows : MIL_TILE;
    const uint32_t units = (rows + g - 1) / g;
    cuts.assign(1, 0u);
    for (uint32_t k = 1; k <= n; k++) {
        uint64_t c = ((uint64_t)units * k / n) * g;
        if (c > rows || k == n) c = rows;
        if (c > cuts.back()) cuts.push_back((uint32_t)c);
    }
    if (cuts.back() != rows) cuts.push_back(rows);
}
// the parts of `pieces` that fall into local rows [k0, k1)
void pieces_in(const std::vector<Piece> &pieces, uint32_t k0, uint32_t k1, std::vector<Piece> &out)
{
    out.clear();
    for (const Piece &p : pieces) {
        const uint32_t a = p.local0 > k0 ? p.local0 : k0, b = p.local0 + p.rows < k1 ? p.local0 + p.rows : k1;
        if (b > a) out.push_back(Piece{ a, b - a, p.frame0 + (a - p.local0) });
    }
}

size_t bytes_per_px(mil_out_format fmt) { return fmt == MIL_OUT_HDR_RGBA32F ? 16u : 4u; }

// A rank's streams must not share a hardware queue with one another: HIP deals a process's streams over its hardware queues
// (GPU_MAX_HW_QUEUES, 4 here) in creation order; two streams on one queue run their kernels one after the other, and an
// event wait of one of them holds back every later packet of the queue - a second lane would buy nothing there, and a
// gather stream (whose RCCL kernels wait for their peers) would stall the lane it shares a queue with.  Probe: a 300 us spin
// kernel (one wave) on every stream of `against` and on a fresh candidate at once; if the lot takes about one spin they run
// side by side.  Up to four candidates, the rest are destroyed; if none is free of the others the last one is kept (correct
// either way: the order of frames rests on events).  ~1 ms per stream, once per rank at creation.
int pick_stream(mil_mg *m, Local &l, hipStream_t *against, int n_against, const char *what, hipStream_t *out, bool *side_by_side)
{
    const uint32_t spin_us = 300;
    MG_HIP(m, launch_delay(against[0], 1));               // the spin kernel's code object is loaded outside the timed runs
    MG_HIP(m, hipStreamSynchronize(against[0]));
    const int max_tries = 4;
    for (int tries = 0; tries < max_tries; tries++) {
        hipStream_t cand = nullptr;
        MG_HIP(m, hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
        for (int k = 0; k < n_against; k++) MG_HIP(m, hipStreamSynchronize(against[k]));
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        for (int k = 0; k < n_against && e == hipSuccess; k++) e = launch_delay(against[k], spin_us);
        if (e == hipSuccess) e = launch_delay(cand, spin_us);
        for (int k = 0; k < n_against && e == hipSuccess; k++) e = hipStreamSynchronize(against[k]);
        if (e == hipSuccess) e = hipStreamSynchronize(cand);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (e != hipSuccess) { (void)hipStreamDestroy(cand); return hipf(m, e); }
        const bool ok = us < 1.6 * spin_us;
        if (ok || tries == max_tries - 1) {
            *out = cand;
            *side_by_side = ok;
            if (getenv("MIL_DEBUG")) fprintf(stderr, "mil_mg rank %d: %s stream after %d candidate(s), %d spins of %u us took %.0f us (%s)\n",
                                             l.rank, what, tries + 1, n_against + 1, spin_us, us, ok ? "side by side" : "one after the other");
            return MIL_OK;
        }
        (void)hipStreamDestroy(cand);
    }
    return MIL_OK;
}

int create_common(mil_mg *m, const mil_mg_config *cfg)
{
    for (Local &l : m->locals) {
        const char *env = getenv("MIL_MG_LANES");
        l.lanes = (env && atoi(env) == 1) ? 1 : Local::LANES;
        for (int k = 0; k < l.lanes; k++) {
            const int rc = mil_create(&l.ctx[k], l.device, cfg->max_point_lights);
            if (rc) return rc;
        }
        MG_HIP(m, hipSetDevice(l.device));
        // One compute stream per lane; a frame's chunks all run on its lane's stream.  (Measured, 8K / 4096 lights, one eighth
        // of the frame: 405 us as one launch, 553 as two launches on one stream - the tail of the heavy tiles is paid per
        // launch - and 461 with the first chunk on a higher-priority stream beside the second; but with 8 ranks x 3 streams on
        // the 4 hardware queues of this pool (GPU_MAX_HW_QUEUES) mixed-priority streams stopped making progress: all streams
        // here have the default priority.)
        MG_HIP(m, hipStreamCreateWithFlags(&l.compute[0], hipStreamNonBlocking));
        if (l.lanes > 1) {
            const int rc = pick_stream(m, l, l.compute, 1, "second lane", &l.compute[1], &l.lanes_overlap);
            if (rc) return rc;
        }
        {
            bool free_of_lanes = false;
            const int rc = pick_stream(m, l, l.compute, l.lanes, "gather", &l.gather, &free_of_lanes);
            if (rc) return rc;
        }
        for (hipEvent_t &e : l.gathered) MG_HIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const uint32_t n = cfg->chunks ? cfg->chunks : 4u;
        l.chunk_done.resize(n + 1);
        for (hipEvent_t &e : l.chunk_done) MG_HIP(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return MIL_OK;
}

// The staging slices on rank 0's device hold `slice` bytes per rank.  Grows only; growing waits for every transfer of every
// local rank (peer copies into the old buffer run on the senders' gather streams) and allocates: mil_mg_reserve does it
// ahead of the frame loop.
int stage_reserve(mil_mg *m, size_t slice)
{
    if (slice <= m->stage_slice) return MIL_OK;
    const Local *root = nullptr;
    for (const Local &l : m->locals) if (l.rank == 0) root = &l;
    if (!root) return MIL_OK;
    for (Local &l : m->locals) {
        MG_HIP(m, hipSetDevice(l.device));
        MG_HIP(m, hipStreamSynchronize(l.gather));
    }
    MG_HIP(m, hipSetDevice(root->device));
    if (m->stage) { MG_HIP(m, hipFree(m->stage)); m->stage = nullptr; m->stage_slice = 0; }
    MG_HIP(m, hipMalloc((void **)&m->stage, slice * m->world));
    m->stage_slice = slice;
    m->stage_free_valid = false;
    if (!m->stage_free) MG_HIP(m, hipEventCreateWithFlags(&m->stage_free, hipEventDisableTiming));
    return MIL_OK;
}

// One frame: every local rank shades its rows in chunks; after each chunk the rows go to rank 0.
//   dst_of(rank)  base of the image on rank 0's device that receives that rank's rows, indexed by FRAME rows (strips: the
//                 assembled frame for every rank) or by the rank's own rows (frames: the rank's slot)
//   frames        one FrameMatrixData for all ranks (strips) or one per local rank (frames)
int render(mil_mg *m, const mil_mg_rank_io *io, const mil_frame_matrix_data *frames, bool per_rank_frame,
           const mil_directional_light *lights, const mil_push_constants *push, mil_out_format fmt, uint32_t flags,
           void *const *dst_of, uint32_t dst_pitch_px, bool strips)
{
    if (!m || !io || !frames || !push || !dst_of) return MIL_ERR_BAD_ARG;
    // the resolution is validated as a float (finite, integral, 1 .. 2^24) BEFORE it is converted: a NaN, negative or huge
    // value makes the conversion itself undefined
    for (int a = 0; a < 2; a++) {
        const float v = push->resolution[a];
        if (!(v >= 1.0f) || !(v <= 16777216.0f) || v != __builtin_floorf(v)) return MIL_ERR_BAD_ARG;
    }
    const uint32_t height = (uint32_t)push->resolution[1];
    const size_t bpp = bytes_per_px(fmt);
    const bool loopback = (m->cfg.flags & MIL_MG_FLAG_LOOPBACK) != 0;
    const bool use_rccl = m->cfg.transport == MIL_MG_TRANSPORT_RCCL;
    const mil_mg_config layout_cfg = strips ? m->cfg : mil_mg_config{ m->cfg.max_point_lights, 0u, m->cfg.chunks, m->cfg.transport, m->cfg.flags };
    const int layout_world = strips ? m->world : 1;          // frames: every rank holds a whole frame

    // ---- the layout of EVERY rank (the ranks of other processes too: both ends of a transfer derive it the same way)
    struct Plan { std::vector<Piece> pieces; std::vector<uint32_t> cuts; uint32_t rows = 0; };
    std::vector<Plan> plan(m->world);
    size_t n_chunks = 0;
    for (int r = 0; r < m->world; r++) {
        Plan &pl = plan[r];
        rank_pieces(layout_cfg, layout_world, strips ? r : 0, height, pl.pieces);
        pl.rows = pl.pieces.empty() ? 0u : pl.pieces.back().local0 + pl.pieces.back().rows;
        if (pl.rows) chunk_cuts(m->cfg, pl.rows, pl.cuts);        // else: more ranks than bands, nothing to shade there
        if (pl.cuts.size() > n_chunks + 1) n_chunks = pl.cuts.size() - 1;
    }
    // ---- validate everything before anything is enqueued
    const Local *root = nullptr;
    for (const Local &l : m->locals) if (l.rank == 0) root = &l;
    if (root)
        for (int r = 0; r < m->world; r++)
            if (!dst_of[r]) return MIL_ERR_BAD_ARG;
    std::vector<bool> in_place(m->locals.size(), false);
    for (size_t i = 0; i < m->locals.size(); i++) {
        const Local &l = m->locals[i];
        const Plan &pl = plan[l.rank];
        if (pl.rows == 0) continue;
        const mil_mg_rank_io &r = io[i];
        if (r.gb.height != pl.rows || r.gb.width == 0 || (float)r.gb.width != push->resolution[0]) return MIL_ERR_BAD_ARG;
        if (r.extras && r.extras->ldr_out) return MIL_ERR_BAD_ARG;      // one gathered image per call: shade to hdr, or to a packed format
        // rank 0 shades straight into its destination unless the loopback test mode routes it through the transport
        in_place[i] = l.rank == 0 && !loopback;
        if (!in_place[i] && (!r.out || r.out_pitch_px != r.gb.width)) return MIL_ERR_BAD_ARG;     // gathered rows travel as dense runs
        if (dst_pitch_px != r.gb.width && !(in_place[i] && dst_pitch_px > r.gb.width && m->world == 1)) return MIL_ERR_BAD_ARG;
    }

    // ---- dealt bands travel as ONE message per rank and chunk (the sender's rows are contiguous) into a staging slice on
    // rank 0's device and are scattered into the frame there by a strided copy: 34 messages of 491 KB per rank and 8K frame
    // (16-row bands) would each pay the point-to-point set-up on the one channel a peer pair uses.  Contiguous strips and
    // whole frames have one piece and are received in place.
    uint32_t max_rows = 0;
    for (int r = 0; r < m->world; r++) max_rows = plan[r].rows > max_rows ? plan[r].rows : max_rows;
    const size_t row_bytes = (size_t)push->resolution[0] * bpp;
    const bool staged = strips && m->cfg.band_rows != 0;
    if (staged && root) {
        const int rc = stage_reserve(m, (size_t)max_rows * row_bytes);      // a no-op after mil_mg_reserve for this frame size
        if (rc) return rc;
    }

    // ---- enqueue, chunk by chunk
    std::vector<Piece> part;
    struct Scatter { int rank; uint32_t k0, k1; };
    std::vector<Scatter> scatter;
    // an earlier frame's transfers must have left the buffers this frame writes (see Local::touched)
    for (size_t i = 0; i < m->locals.size(); i++) {
        Local &l = m->locals[i];
        std::vector<const void *> mine;
        if (io[i].out) mine.push_back(io[i].out);
        if (l.rank == 0) for (int r = 0; r < m->world; r++) mine.push_back(dst_of[r]);
        MG_HIP(m, hipSetDevice(l.device));
        const int slot = l.frame_next;
        for (int k = 0; k < Local::FRAMES_IN_FLIGHT; k++) {
            if (!l.gathered_valid[k]) continue;
            bool hit = k == slot;                                    // the slot about to be reused: its frame must be done
            for (const void *a : mine) for (const void *b : l.touched[k]) hit = hit || a == b;
            if (hit) {
                MG_HIP(m, hipStreamWaitEvent(l.compute[l.frames % l.lanes], l.gathered[k], 0));
                if (l.rank == 0) MG_HIP(m, hipStreamWaitEvent(l.gather, l.gathered[k], 0));
            }
        }
        l.touched[slot] = mine;
    }
    for (size_t c = 0; c < n_chunks; c++) {
        // shade chunk c on every local rank
        for (size_t i = 0; i < m->locals.size(); i++) {
            Local &l = m->locals[i];
            const Plan &pl = plan[l.rank];
            if (c + 1 >= pl.cuts.size()) continue;
            const uint32_t k0 = pl.cuts[c], k1 = pl.cuts[c + 1];
            const mil_mg_rank_io &r = io[i];
            mil_gbuffer gb = r.gb;
            gb.normal_l_model += (size_t)k0 * gb.pitch_px; gb.albedo_occlusion += (size_t)k0 * gb.pitch_px;
            gb.roughness_metallic += (size_t)k0 * gb.pitch_px; gb.depth += (size_t)k0 * gb.pitch_px;
            gb.height = k1 - k0;
            const mil_frame_matrix_data *f = per_rank_frame ? &frames[i] : frames;
            const uint32_t out_pitch = in_place[i] ? dst_pitch_px : r.out_pitch_px;
            int rc;
            const int lane = (int)(l.frames % l.lanes);
            hipStream_t cs = l.compute[lane];
            if (strips && m->cfg.band_rows) {
                const mil_band_layout b = { m->cfg.band_rows, (uint32_t)m->world, (uint32_t)l.rank, k0 / m->cfg.band_rows };
                gb.y_offset = 0;
                void *out = in_place[i] ? dst_of[0] : (void *)((uint8_t *)r.out + (size_t)k0 * out_pitch * bpp);
                rc = shade(l.ctx[lane], &gb, f, lights, push, r.extras, nullptr, out, out_pitch, fmt, flags, cs, &b, in_place[i], nullptr);
            } else {
                const uint32_t y0 = pl.pieces[0].frame0 + k0;      // contiguous rows: one piece
                gb.y_offset = y0;
                void *out = in_place[i] ? (void *)((uint8_t *)dst_of[0] + (size_t)(strips ? y0 : k0) * out_pitch * bpp)
                                        : (void *)((uint8_t *)r.out + (size_t)k0 * out_pitch * bpp);
                rc = shade(l.ctx[lane], &gb, f, lights, push, r.extras, nullptr, out, out_pitch, fmt, flags, cs, nullptr, false, nullptr);
            }
            if (rc) return rc;
            MG_HIP(m, hipSetDevice(l.device));
            MG_HIP(m, hipEventRecord(l.chunk_done[c], cs));
            MG_HIP(m, hipStreamWaitEvent(l.gather, l.chunk_done[c], 0));
        }
        // gather chunk c: senders -> rank 0.  One group holds every send and receive of this process (RCCL requires that of a
        // thread driving several devices); pairs are matched by their order per peer, which both sides derive from the layout.
        if (use_rccl) MG_NCCL(m, rccl().GroupStart());
        int err = MIL_OK;
        for (int r = 0; r < m->world; r++) {
            const Local *src = nullptr;
            size_t si = 0;
            for (size_t i = 0; i < m->locals.size(); i++) if (m->locals[i].rank == r) { src = &m->locals[i]; si = i; }
            if (r == 0 && !loopback) continue;                          // rank 0 wrote in place
            if (!src && !root) continue;                                // neither end lives in this process
            const Plan &pl = plan[r];
            if (c + 1 >= pl.cuts.size()) continue;
            if (staged) {                                               // one message, scattered on rank 0 after the group
                const uint32_t k0 = pl.cuts[c], k1 = pl.cuts[c + 1];
                const size_t n = (size_t)(k1 - k0) * row_bytes;
                uint8_t *d = root ? m->stage + (size_t)r * m->stage_slice + (size_t)k0 * row_bytes : nullptr;
                const uint8_t *sp = src ? (const uint8_t *)io[si].out + (size_t)k0 * row_bytes : nullptr;
                if (err != MIL_OK) continue;
                if (use_rccl) {
                    if (src) { const ncclResult_t e = rccl().Send(sp, n, ncclChar, 0, src->comm, src->gather); if (e != ncclSuccess) err = ncclf(m, e); }
                    if (root && err == MIL_OK) { const ncclResult_t e = rccl().Recv(d, n, ncclChar, r, root->comm, root->gather); if (e != ncclSuccess) err = ncclf(m, e); }
                } else if (src && root) {
                    // the emulation's sender stream carries the copy; the scatter below runs on rank 0's gather stream behind it
                    hipError_t e = hipSetDevice(src->device);
                    // the slice may still hold the previous frame's rows, not yet scattered by rank 0's gather stream
                    if (e == hipSuccess && m->stage_free_valid && src != root) e = hipStreamWaitEvent(src->gather, m->stage_free, 0);
                    if (e == hipSuccess) e = hipMemcpyAsync(d, sp, n, hipMemcpyDeviceToDevice, src->gather);
                    if (e == hipSuccess) e = hipEventRecord(src->chunk_done[m->locals[si].chunk_done.size() - 1], src->gather);
                    if (e == hipSuccess) e = hipStreamWaitEvent(root->gather, src->chunk_done[m->locals[si].chunk_done.size() - 1], 0);
                    if (e != hipSuccess) err = hipf(m, e);
                }
                if (root && err == MIL_OK) scatter.push_back(Scatter{ r, k0, k1 });
                continue;
            }
            pieces_in(pl.pieces, pl.cuts[c], pl.cuts[c + 1], part);
            for (const Piece &p : part) {
                const size_t n = (size_t)p.rows * row_bytes;
                uint8_t *d = root ? (uint8_t *)dst_of[r] + (size_t)(strips ? p.frame0 : p.local0) * row_bytes : nullptr;
                const uint8_t *sp = src ? (const uint8_t *)io[si].out + (size_t)p.local0 * row_bytes : nullptr;
                if (err != MIL_OK) break;
                if (use_rccl) {
                    if (src) { const ncclResult_t e = rccl().Send(sp, n, ncclChar, 0, src->comm, src->gather); if (e != ncclSuccess) err = ncclf(m, e); }
                    if (root && err == MIL_OK) { const ncclResult_t e = rccl().Recv(d, n, ncclChar, r, root->comm, root->gather); if (e != ncclSuccess) err = ncclf(m, e); }
                } else if (src && root) {                               // peer copy on the sender's gather stream (one process)
                    hipError_t e = hipSetDevice(src->device);
                    if (e == hipSuccess) e = hipMemcpyAsync(d, sp, n, hipMemcpyDeviceToDevice, src->gather);
                    if (e != hipSuccess) err = hipf(m, e);
                }
            }
        }
        if (use_rccl) {
            const ncclResult_t e = rccl().GroupEnd();
            if (e != ncclSuccess && err == MIL_OK) err = ncclf(m, e);
        }
        if (err) return err;
        // staging slice -> frame rows, on rank 0's gather stream behind the receives: runs of whole bands as one strided copy
        if (!scatter.empty()) MG_HIP(m, hipSetDevice(root->device));
        for (const Scatter &sc : scatter) {
            pieces_in(plan[sc.rank].pieces, sc.k0, sc.k1, part);
            const uint8_t *base = m->stage + (size_t)sc.rank * m->stage_slice;
            const uint32_t B = m->cfg.band_rows, stride = B * (uint32_t)m->world;
            size_t i = 0;
            while (i < part.size()) {
                size_t j = i + 1;
                if (part[i].rows == B)
                    while (j < part.size() && part[j].rows == B && part[j].frame0 == part[j - 1].frame0 + stride && part[j].local0 == part[j - 1].local0 + B) j++;
                uint8_t *d = (uint8_t *)dst_of[sc.rank] + (size_t)part[i].frame0 * row_bytes;
                const uint8_t *sp = base + (size_t)part[i].local0 * row_bytes;
                if (j - i > 1) MG_HIP(m, hipMemcpy2DAsync(d, (size_t)stride * row_bytes, sp, (size_t)B * row_bytes, (size_t)B * row_bytes, j - i, hipMemcpyDeviceToDevice, root->gather));
                else MG_HIP(m, hipMemcpyAsync(d, sp, (size_t)part[i].rows * row_bytes, hipMemcpyDeviceToDevice, root->gather));
                i = j;
            }
        }
        scatter.clear();
    }
    if (staged && root && m->stage_free) {          // the slices are free again once this frame's scatters have run
        MG_HIP(m, hipSetDevice(root->device));
        MG_HIP(m, hipEventRecord(m->stage_free, root->gather));
        m->stage_free_valid = true;
    }
    for (Local &l : m->locals) {
        MG_HIP(m, hipSetDevice(l.device));
        MG_HIP(m, hipEventRecord(l.gathered[l.frame_next], l.gather));
        l.gathered_valid[l.frame_next] = true;
        l.frame_next = (l.frame_next + 1) % Local::FRAMES_IN_FLIGHT;
        l.frames++;
    }
    return MIL_OK;
}

} // namespace

extern "C" {

int mil_mg_unique_id(void *id)
{
    if (!id) return MIL_ERR_BAD_ARG;
    if (!rccl().ok) return MIL_ERR_RCCL;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return MIL_ERR_RCCL;
    memcpy(id, &u, sizeof(u));
    return MIL_OK;
}

void mil_mg_destroy(mil_mg *m)
{
    if (!m) return;
    for (Local &l : m->locals) {
        (void)hipSetDevice(l.device);
        for (hipStream_t cs : l.compute) if (cs) (void)hipStreamSynchronize(cs);
        if (l.gather) (void)hipStreamSynchronize(l.gather);
    }
    for (Local &l : m->locals) {
        (void)hipSetDevice(l.device);
        if (l.comm && rccl().ok) (void)rccl().CommDestroy(l.comm);
        for (hipEvent_t e : l.chunk_done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : l.gathered) if (e) (void)hipEventDestroy(e);
        for (hipStream_t cs : l.compute) if (cs) (void)hipStreamDestroy(cs);
        if (l.gather) (void)hipStreamDestroy(l.gather);
        for (mil_ctx *c : l.ctx) if (c) mil_destroy(c);
    }
    if (m->stage || m->stage_free) {
        for (Local &l : m->locals) if (l.rank == 0) (void)hipSetDevice(l.device);
        if (m->stage) (void)hipFree(m->stage);
        if (m->stage_free) (void)hipEventDestroy(m->stage_free);
    }
    delete m;
}

int mil_mg_create(mil_mg **out, int n_devices, const int *devices, const mil_mg_config *cfg)
{
    if (!out) return MIL_ERR_BAD_ARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !devices || !cfg || !cfg_ok(cfg)) return MIL_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return MIL_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= n) return MIL_ERR_NO_DEVICE;
    const bool use_rccl = cfg->transport == MIL_MG_TRANSPORT_RCCL;
    if (use_rccl && !rccl().ok) return MIL_ERR_RCCL;
    mil_mg *m = new (std::nothrow) mil_mg();
    if (!m) return MIL_ERR_HIP;
    m->world = n_devices; m->cfg = *cfg; m->single_process = true;
    m->locals.resize(n_devices);
    for (int i = 0; i < n_devices; i++) { m->locals[i].rank = i; m->locals[i].device = devices[i]; }
    int rc = create_common(m, cfg);
    if (rc == MIL_OK && use_rccl) {
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t e = rccl().CommInitAll(comms.data(), n_devices, devices);     // refuses two ranks on one device
        if (e != ncclSuccess) rc = ncclf(m, e);
        else {
            for (int i = 0; i < n_devices; i++) m->locals[i].comm = comms[i];
            (void)rccl().CommCount(comms[0], &m->rccl_ranks);
        }
    }
    if (rc == MIL_OK && !use_rccl)                                                       // peer copies need the mappings
        for (int i = 1; i < n_devices; i++)
            if (devices[i] != devices[0]) {
                (void)hipSetDevice(devices[i]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { rc = hipf(m, e); break; }
                (void)hipGetLastError();
            }
    if (rc != MIL_OK) {
        mil_mg_destroy(m);
        return rc;
    }
    *out = m;
    return MIL_OK;
}

int mil_mg_create_rank(mil_mg **out, int world, int rank, int device, const void *id, const mil_mg_config *cfg)
{
    if (!out) return MIL_ERR_BAD_ARG;
    *out = nullptr;
    if (world < 1 || world > 4096 || rank < 0 || rank >= world || !id || !cfg || !cfg_ok(cfg)) return MIL_ERR_BAD_ARG;
    if (cfg->transport != MIL_MG_TRANSPORT_RCCL) return MIL_ERR_BAD_ARG;               // peer copies need one process
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MIL_ERR_NO_DEVICE;
    if (!rccl().ok) return MIL_ERR_RCCL;
    mil_mg *m = new (std::nothrow) mil_mg();
    if (!m) return MIL_ERR_HIP;
    m->world = world; m->cfg = *cfg; m->single_process = false;
    m->locals.resize(1);
    m->locals[0].rank = rank; m->locals[0].device = device;
    int rc = create_common(m, cfg);
    if (rc == MIL_OK) {
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        (void)hipSetDevice(device);
        const ncclResult_t e = rccl().CommInitRank(&m->locals[0].comm, world, u, rank);
        if (e != ncclSuccess) rc = ncclf(m, e);
        else (void)rccl().CommCount(m->locals[0].comm, &m->rccl_ranks);
    }
    if (rc != MIL_OK) { mil_mg_destroy(m); return rc; }
    *out = m;
    return MIL_OK;
}

int mil_mg_world(const mil_mg *m) { return m ? m->world : 0; }
int mil_mg_local_ranks(const mil_mg *m) { return m ? (int)m->locals.size() : 0; }
int mil_mg_rccl_ranks(const mil_mg *m) { return m ? m->rccl_ranks : 0; }
int mil_mg_last_rccl_error(const mil_mg *m) { return m ? m->rccl_error : 0; }
int mil_mg_last_hip_error(const mil_mg *m) { return m ? m->last_hip : 0; }

uint32_t mil_mg_rows_of_rank(const mil_mg_config *cfg, int world, int rank, uint32_t height)
{
    if (!cfg || !cfg_ok(cfg) || world < 1 || rank < 0 || rank >= world) return 0;
    return rank_rows(*cfg, world, rank, height);
}

int mil_mg_frame_rows_of_rank(const mil_mg_config *cfg, int world, int rank, uint32_t height, uint32_t *frame_rows)
{
    if (!cfg || !cfg_ok(cfg) || world < 1 || rank < 0 || rank >= world || !frame_rows) return MIL_ERR_BAD_ARG;
    std::vector<Piece> p;
    rank_pieces(*cfg, world, rank, height, p);
    for (const Piece &q : p)
        for (uint32_t k = 0; k < q.rows; k++) frame_rows[q.local0 + k] = q.frame0 + k;
    return MIL_OK;
}

int mil_mg_render_strips(mil_mg *m, const mil_mg_rank_io *io, const mil_frame_matrix_data *frame,
                         const mil_directional_light *lights, const mil_push_constants *push, mil_out_format fmt,
                         uint32_t flags, void *full, uint32_t full_pitch_px)
{
    if (!m) return MIL_ERR_BAD_ARG;
    std::vector<void *> dst(m->world, full);            // every rank's rows land in the one assembled frame
    return render(m, io, frame, false, lights, push, fmt, flags, dst.data(), full_pitch_px, true);
}

int mil_mg_render_frames(mil_mg *m, const mil_mg_rank_io *io, const mil_frame_matrix_data *frames,
                         const mil_directional_light *lights, const mil_push_constants *push, mil_out_format fmt,
                         uint32_t flags, void *const *slots, uint32_t slot_pitch_px)
{
    if (!m || !slots) return MIL_ERR_BAD_ARG;
    return render(m, io, frames, true, lights, push, fmt, flags, slots, slot_pitch_px, false);
}

int mil_mg_reserve(mil_mg *m, uint32_t width, uint32_t height, mil_out_format fmt)
{
    if (!m || width == 0 || height == 0 || width > (1u << 24) || height > (1u << 24) || (unsigned)fmt > MIL_OUT_SRGB8_BGRA_VIA_LDR) return MIL_ERR_BAD_ARG;
    // every local rank's context: the two-kernel workspace for the rows it will shade (strips: its share; frames: a whole frame)
    for (Local &l : m->locals) {
        for (int k = 0; k < l.lanes; k++) {
            const int rc = mil_reserve(l.ctx[k], width, height);
            if (rc) return rc;
        }
    }
    if (m->cfg.band_rows == 0) return MIL_OK;          // contiguous strips and whole frames are received in place
    uint32_t max_rows = 0;
    for (int r = 0; r < m->world; r++) {
        const uint32_t rows = rank_rows(m->cfg, m->world, r, height);
        max_rows = rows > max_rows ? rows : max_rows;
    }
    return stage_reserve(m, (size_t)max_rows * width * bytes_per_px(fmt));
}

int mil_mg_debug_delay_gather(mil_mg *m, uint32_t microseconds)
{
    if (!m) return MIL_ERR_BAD_ARG;
    for (Local &l : m->locals)
        if (l.rank == 0) {
            MG_HIP(m, hipSetDevice(l.device));
            MG_HIP(m, launch_delay(l.gather, microseconds));
        }
    return MIL_OK;
}

int mil_mg_debug_delay_senders(mil_mg *m, uint32_t microseconds)
{
    if (!m) return MIL_ERR_BAD_ARG;
    for (Local &l : m->locals)
        if (l.rank != 0) {
            MG_HIP(m, hipSetDevice(l.device));
            MG_HIP(m, launch_delay(l.gather, microseconds));
        }
    return MIL_OK;
}

int mil_mg_sync(mil_mg *m)
{
    if (!m) return MIL_ERR_BAD_ARG;
    for (Local &l : m->locals) {
        MG_HIP(m, hipSetDevice(l.device));
        for (int k = 0; k < l.lanes; k++) MG_HIP(m, hipStreamSynchronize(l.compute[k]));
        MG_HIP(m, hipStreamSynchronize(l.gather));
        if (l.comm && rccl().ok) {
            ncclResult_t ae = ncclSuccess;
            MG_NCCL(m, rccl().CommGetAsyncError(l.comm, &ae));
            if (ae != ncclSuccess) return ncclf(m, ae);
        }
    }
    return MIL_OK;
}

} // extern "C"
