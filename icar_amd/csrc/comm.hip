// icar_amd/csrc/comm.hip -- row H1 transport and co_min behind the C ABI.
//
// The reference moves halos with coarray PUTs into the neighbour's halo_*_in buffers followed by `sync images`
// (src/objects/exchangeable_obj.f90:138-356, one PUT per variable and direction) and reduces the time step with
// `call co_min(seconds)` (src/main/time_step.f90:413).  Here ONE message per neighbour carries every exchanged scalar:
//
//   halo_send      pack kernel (all directions, one launch) -> ncclSend / ncclRecv per neighbour, grouped, on the context's
//                  stream: RCCL over xGMI.  Nothing waits on the host; the interior microphysics runs beside it on the
//                  context's second stream (time_step.f90:512-526).
//   halo_retrieve  unpack kernel (all directions, one launch, the reference's N, S, E, W precedence at the corners);
//                  stream order IS the `sync images`.
//   co_min         ncclAllReduce of one value on the device.
//
// Edges that wrap around to the tile itself (a periodic single image, what src/tests/test_mpdata.f90 does by hand) use the
// same pack / unpack kernels and no transport.  A second transport, host-staged through POSIX shared memory, exists for boxes
// with fewer GPUs than images (RCCL refuses two ranks on one device): the same entry points, the same kernels, the
// messages PUT into the neighbour's inbox by the CPU.  It is a functional path (tests, 1-GPU boxes), never a benchmark.
//
// librccl.so.1 is opened on the first icar_hip_comm_init with more than a local topology (573 MB; single-image users of
// the library never load it).
#include "ctx.h"
#include "comm.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <thread>

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.so) return 0;
    // a process that already has an RCCL mapped (PyTorch ships its own) gets that one: same soname
    void *so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) { icar_set_error(std::string("comm_init: cannot load librccl.so.1: ") + dlerror()); return 1; }
    Rccl r; r.so = so;
#define SYM(field, name) do { *(void **)(&r.field) = dlsym(so, name); if (!r.field) { icar_set_error("comm_init: librccl lacks " name); dlclose(so); return 1; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString"); SYM(CommCount, "ncclCommCount");
#undef SYM
    g_rccl = r;
    return 0;
}

int rccl_check(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return 0;
    icar_set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
    return 1;
}
#define NCHK(x) do { if (rccl_check((x), #x)) return 1; } while (0)

// ---- shared-memory segment of the host-staged transport -------------------------------------------------------------
struct alignas(64) ShmBox { std::atomic<uint64_t> seq; std::atomic<uint64_t> ack; };
struct alignas(64) ShmRed { std::atomic<uint64_t> seq; double val[2]; };
struct ShmHeader { uint64_t magic, nranks, slot_bytes; };
constexpr uint64_t kMagic = 0x4943415248495031ull;          // "ICARHIP1"
double g_wait_seconds = 60.0;                                // a lost neighbour is an error, not a hang (icar_hip_comm_timeout)

inline int opposite(int d) { return d ^ 1; }                 // north 0 <-> south 1, east 2 <-> west 3

__global__ void k_stamp_fill(float *__restrict__ f, size_t n, float v)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) f[t] = v;
}
// direction order of the transport: north 0 (rows ny-h..), south 1 (rows ..h-1), east 2 (columns nx-h..), west 3 (columns ..h-1);
// want < 0: no neighbour on that side, its halo cells keep this image's own stamp and are not checked
__global__ void k_stamp_check(Dims d, const float *__restrict__ f, int h, float wn, float ws, float we, float ww, int *__restrict__ bad)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)d.nx * d.nz * d.ny) return;
    const int i = (int)(t % d.nx), j = (int)(t / ((size_t)d.nx * d.nz));
    const bool n_ = j >= d.ny - h, s_ = j < h, e_ = i >= d.nx - h, w_ = i < h;
    if ((n_ || s_) && (e_ || w_)) return;                      // corner
    const float want = n_ ? wn : s_ ? ws : e_ ? we : w_ ? ww : -1.0f;
    if (want >= 0.0f && f[t] != want) atomicAdd(bad, 1);
}

}  // namespace

struct IcarComm {
    int kind = ICAR_COMM_LOCAL;
    int nranks = 1, rank = 0;
    int nb[4] = {ICAR_NEIGHBOR_NONE, ICAR_NEIGHBOR_NONE, ICAR_NEIGHBOR_NONE, ICAR_NEIGHBOR_NONE};
    float *sbuf[4] = {nullptr, nullptr, nullptr, nullptr}, *rbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};                            // elements allocated per direction
    size_t cnt[4] = {0, 0, 0, 0};                            // elements of the message in flight per direction
    bool in_flight = false;
    // RCCL
    ncclComm_t nccl = nullptr;
    double *d_red = nullptr;                                 // 8 bytes on the device for the scalar reductions
    double *h_red = nullptr;                                 // pinned
    // host-staged
    std::string shm_name; void *shm = nullptr; size_t shm_bytes = 0, slot_bytes = 0;
    float *hs[4] = {nullptr, nullptr, nullptr, nullptr}, *hr[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging
    uint64_t msg_no = 0, red_no = 0;
    // exchange_u / exchange_v messages (one per neighbour, both fields): device send / receive buffers, pinned staging
    float *uvs[4] = {nullptr, nullptr, nullptr, nullptr}, *uvr[4] = {nullptr, nullptr, nullptr, nullptr};
    float *uvhs[4] = {nullptr, nullptr, nullptr, nullptr}, *uvhr[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t uvcap_s[4] = {0, 0, 0, 0}, uvcap_r[4] = {0, 0, 0, 0};

    ShmBox *box(int r, int d) const { return (ShmBox *)((char *)shm + 64 + ((size_t)r * 5 + d) * 64); }
    ShmRed *red(int r) const { return (ShmRed *)((char *)shm + 64 + ((size_t)r * 5 + 4) * 64); }
    char *slot(int r, int d) const { return (char *)shm + 64 + (size_t)nranks * 5 * 64 + ((size_t)r * 4 + d) * slot_bytes; }
};

static bool has_peer(const IcarComm *m, int d) { return m->nb[d] >= 0; }
static bool wraps(const IcarComm *m, int d) { return m->nb[d] == ICAR_NEIGHBOR_SELF; }

static int ensure_buffers(icar_hip_ctx *c, IcarComm *m, int h, int nf)
{
    for (int d = 0; d < 4; ++d) {
        if (m->nb[d] == ICAR_NEIGHBOR_NONE) { m->cnt[d] = 0; continue; }
        const size_t n = icar_hip_halo_count(c, d, h) * (size_t)nf;
        m->cnt[d] = n;
        if (n <= m->cap[d]) continue;
        if (m->sbuf[d]) { hipFree(m->sbuf[d]); m->sbuf[d] = nullptr; }
        if (m->rbuf[d]) { hipFree(m->rbuf[d]); m->rbuf[d] = nullptr; }
        m->cap[d] = 0;                                       // nothing usable until every allocation below has succeeded
        HIPCHK(hipMalloc(&m->sbuf[d], n * sizeof(float)));
        if (has_peer(m, d)) HIPCHK(hipMalloc(&m->rbuf[d], n * sizeof(float)));
        if (m->kind == ICAR_COMM_HOST && has_peer(m, d)) {
            if (m->hs[d]) hipHostFree(m->hs[d]);
            if (m->hr[d]) hipHostFree(m->hr[d]);
            HIPCHK(hipHostMalloc((void **)&m->hs[d], n * sizeof(float), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&m->hr[d], n * sizeof(float), hipHostMallocDefault));
        }
        m->cap[d] = n;
    }
    return 0;
}

template <class Pred>
static int spin_until(Pred p, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0; !p(); ++it) {
        if ((it & 1023) == 1023) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g_wait_seconds) {
                icar_set_error(std::string(what) + ": no answer from a neighbouring image within the wait limit (icar_hip_comm_timeout)"); return 1;
            }
            std::this_thread::yield();
        }
    }
    return 0;
}

// ---- halo_send / halo_retrieve (domain_obj.f90:109-143) --------------------------------------------------------------
int icar_comm_halo_send(icar_hip_ctx *c, int h, const int *fields, int nf)
{
    IcarComm *m = c->comm;
    if (!m || nf <= 0) return 0;
    if (m->in_flight) { icar_set_error("halo_send: the previous halo_send has not been retrieved"); return 1; }
    if (ensure_buffers(c, m, h, nf)) return 1;
    int dirs[4], nd = 0; void *bufs[4];
    for (int d = 0; d < 4; ++d) if (m->nb[d] != ICAR_NEIGHBOR_NONE) { dirs[nd] = d; bufs[nd] = m->sbuf[d]; ++nd; }
    if (!nd) return 0;
    if (icar_halo_pack_dirs(c, nd, dirs, h, fields, nf, bufs, false)) return 1;      // put_<dir> of every variable, one launch
    bool peers = false;
    for (int d = 0; d < 4; ++d) peers = peers || has_peer(m, d);
    if (peers && m->kind == ICAR_COMM_RCCL) {
        // what I send towards d arrives at the peer as coming from opposite(d).  Sends are posted N, S, E, W and receives in the
        // order of the matching sends of the peer (from S, N, W, E), so two images that are each other's neighbour in both
        // directions of an axis (a periodic pair) pair their messages correctly: RCCL matches per peer in posting order.
        ScopedTimer t(c, "halo_transport");
        NCHK(g_rccl.GroupStart());
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) NCHK(g_rccl.Send(m->sbuf[d], m->cnt[d], ncclFloat, m->nb[d], m->nccl, c->stream));
        for (int d = 0; d < 4; ++d) { const int o = opposite(d); if (has_peer(m, o)) NCHK(g_rccl.Recv(m->rbuf[o], m->cnt[o], ncclFloat, m->nb[o], m->nccl, c->stream)); }
        NCHK(g_rccl.GroupEnd());
    } else if (peers && m->kind == ICAR_COMM_HOST) {
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            if (m->cnt[d] * sizeof(float) > m->slot_bytes) { icar_set_error("halo_send: message larger than the slot_bytes given to icar_hip_comm_init_host"); return 1; }
            HIPCHK(hipMemcpyAsync(m->hs[d], m->sbuf[d], m->cnt[d] * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));                 // the strips + pack; the interior launch on the second stream keeps running
        ++m->msg_no;
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            const int p = m->nb[d], o = opposite(d);
            ShmBox *b = m->box(p, o);
            const uint64_t want = m->msg_no - 1;
            if (spin_until([&] { return b->ack.load(std::memory_order_acquire) >= want; }, "halo_send")) return 1;
            memcpy(m->slot(p, o), m->hs[d], m->cnt[d] * sizeof(float));             // the PUT
            b->seq.store(m->msg_no, std::memory_order_release);
        }
    } else if (peers) { icar_set_error("halo_send: neighbours given but no transport (icar_hip_comm_init with a unique id, or _init_host)"); return 1; }
    m->in_flight = true;
    return 0;
}

int icar_comm_halo_retrieve(icar_hip_ctx *c, int h, const int *fields, int nf)
{
    IcarComm *m = c->comm;
    if (!m || nf <= 0) return 0;
    if (!m->in_flight) { icar_set_error("halo_retrieve without a halo_send"); return 1; }
    m->in_flight = false;
    if (m->kind == ICAR_COMM_HOST) {
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            ShmBox *b = m->box(m->rank, d);
            if (spin_until([&] { return b->seq.load(std::memory_order_acquire) >= m->msg_no; }, "halo_retrieve")) return 1;   // sync images
            memcpy(m->hr[d], m->slot(m->rank, d), m->cnt[d] * sizeof(float));
            b->ack.store(m->msg_no, std::memory_order_release);
            HIPCHK(hipMemcpyAsync(m->rbuf[d], m->hr[d], m->cnt[d] * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
    }
    // one unpack launch for everything that arrived: from a peer at d its message (rbuf[d]); at a wrapping edge d what my own
    // opposite edge packed (sbuf[opposite(d)]).  N/S rows leave the corner cells to an E/W message of the same call (the
    // reference retrieves N, S, E, W in that order, exchangeable_obj.f90:138-151).
    int dirs[4], nd = 0; void *bufs[4];
    for (int d = 0; d < 4; ++d) {
        if (has_peer(m, d)) { dirs[nd] = d; bufs[nd] = m->rbuf[d]; ++nd; }
        else if (wraps(m, d) && wraps(m, opposite(d))) { dirs[nd] = d; bufs[nd] = m->sbuf[opposite(d)]; ++nd; }
    }
    if (!nd) return 0;
    return icar_halo_pack_dirs(c, nd, dirs, h, fields, nf, bufs, true);
}

// ---- exchange_u / exchange_v (exchangeable_obj.f90:158-229) ------------------------------------------------------------
// `call domain%u%exchange_u(); call domain%v%exchange_v()` (wind.f90:404-405, :482-483) as ONE message per neighbour: the two
// fields are independent, every box is packed before any is unpacked (the PUTs precede `sync images`), N / S are unpacked
// before E / W like the reference's retrieve order.  A box is (field, i0, ni, j0, nj), 0-based in the field's own memory
// extents (u has nx+1 columns, v has ny+1 rows):
//   u: N / S like exchange (put_north / put_south over the full staggered width, :160-161); the east PUT is halo+1 columns
//      n-2h..n-h (:166) landing in the neighbour's columns start..start+h (:188); the west PUT is columns start+h+1..start+2h
//      (:172) landing in the neighbour's last h columns;   v: the same with rows (:202-221).
namespace {
struct UvBox { int f, i0, ni, j0, nj; };
void uv_plan(int nx, int ny, int h, int dir, UvBox send[2], UvBox recv[2])
{
    const int X = nx + 1, Y = ny + 1, U = ICAR_F_U, V = ICAR_F_V;
    switch (dir) {
    case 0: send[0] = {U, 0, X, ny - 2 * h, h}; send[1] = {V, 0, nx, Y - 1 - 2 * h, h + 1};
            recv[0] = {U, 0, X, ny - h, h};     recv[1] = {V, 0, nx, Y - h, h}; break;
    case 1: send[0] = {U, 0, X, h, h};          send[1] = {V, 0, nx, h + 1, h};
            recv[0] = {U, 0, X, 0, h};          recv[1] = {V, 0, nx, 0, h + 1}; break;
    case 2: send[0] = {U, X - 1 - 2 * h, h + 1, 0, ny}; send[1] = {V, nx - 2 * h, h, 0, Y};
            recv[0] = {U, X - h, h, 0, ny};             recv[1] = {V, nx - h, h, 0, Y}; break;
    default: send[0] = {U, h + 1, h, 0, ny};    send[1] = {V, h, h, 0, Y};
             recv[0] = {U, 0, h + 1, 0, ny};    recv[1] = {V, 0, h, 0, Y}; break;
    }
}
size_t uv_count(const UvBox b[2], int nz) { return ((size_t)b[0].ni * b[0].nj + (size_t)b[1].ni * b[1].nj) * nz; }
}  // namespace

bool icar_comm_has_peers(icar_hip_ctx *c)
{
    if (!c->comm) return false;
    for (int d = 0; d < 4; ++d) if (has_peer(c->comm, d)) return true;
    return false;
}

int icar_comm_exchange_uv(icar_hip_ctx *c, int h, int which)
{
    IcarComm *m = c->comm;
    if (!m) return 0;
    bool peers = false;
    for (int d = 0; d < 4; ++d) peers = peers || has_peer(m, d);
    if (!peers) return 0;                                   // (edges that wrap to the tile itself are not exchanged: the reference has none)
    if (m->in_flight) { icar_set_error("exchange_uv between a halo_send and its halo_retrieve"); return 1; }
    if (m->kind != ICAR_COMM_RCCL && m->kind != ICAR_COMM_HOST) { icar_set_error("exchange_uv: neighbours given but no transport"); return 1; }
    const Dims &dd = c->d;
    if (h < 1 || 2 * h + 1 > dd.nx || 2 * h + 1 > dd.ny) { icar_set_error("exchange_uv: bad halo width"); return 1; }
    UvBox sb[4][2], rb[4][2]; size_t ns[4] = {0, 0, 0, 0}, nr[4] = {0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
        uv_plan(dd.nx, dd.ny, h, d, sb[d], rb[d]);
        ns[d] = uv_count(sb[d], dd.nz); nr[d] = uv_count(rb[d], dd.nz);
        if (ns[d] > m->uvcap_s[d]) {
            if (m->uvs[d]) { (void)hipFree(m->uvs[d]); m->uvs[d] = nullptr; }
            if (m->uvhs[d]) { (void)hipHostFree(m->uvhs[d]); m->uvhs[d] = nullptr; }
            m->uvcap_s[d] = 0;
            HIPCHK(hipMalloc(&m->uvs[d], ns[d] * sizeof(float)));
            if (m->kind == ICAR_COMM_HOST) HIPCHK(hipHostMalloc((void **)&m->uvhs[d], ns[d] * sizeof(float), hipHostMallocDefault));
            m->uvcap_s[d] = ns[d];
        }
        if (nr[d] > m->uvcap_r[d]) {
            if (m->uvr[d]) { (void)hipFree(m->uvr[d]); m->uvr[d] = nullptr; }
            if (m->uvhr[d]) { (void)hipHostFree(m->uvhr[d]); m->uvhr[d] = nullptr; }
            m->uvcap_r[d] = 0;
            HIPCHK(hipMalloc(&m->uvr[d], nr[d] * sizeof(float)));
            if (m->kind == ICAR_COMM_HOST) HIPCHK(hipHostMalloc((void **)&m->uvhr[d], nr[d] * sizeof(float), hipHostMallocDefault));
            m->uvcap_r[d] = nr[d];
        }
        size_t off = 0;
        for (int b = 0; b < 2; ++b) {
            const UvBox &x = sb[d][b];
            if (icar_box_copy(c, x.f, which, x.i0, x.ni, x.j0, x.nj, m->uvs[d] + off, false)) return 1;
            off += (size_t)x.ni * x.nj * dd.nz;
        }
    }
    if (m->kind == ICAR_COMM_RCCL) {
        ScopedTimer t(c, "halo_transport");
        NCHK(g_rccl.GroupStart());
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) NCHK(g_rccl.Send(m->uvs[d], ns[d], ncclFloat, m->nb[d], m->nccl, c->stream));
        for (int d = 0; d < 4; ++d) { const int o = opposite(d); if (has_peer(m, o)) NCHK(g_rccl.Recv(m->uvr[o], nr[o], ncclFloat, m->nb[o], m->nccl, c->stream)); }
        NCHK(g_rccl.GroupEnd());
    } else {
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            if (ns[d] * sizeof(float) > m->slot_bytes) { icar_set_error("exchange_uv: message larger than the slot_bytes given to icar_hip_comm_init_host"); return 1; }
            HIPCHK(hipMemcpyAsync(m->uvhs[d], m->uvs[d], ns[d] * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        ++m->msg_no;
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            const int p = m->nb[d], o = opposite(d);
            ShmBox *b = m->box(p, o);
            const uint64_t want = m->msg_no - 1;
            if (spin_until([&] { return b->ack.load(std::memory_order_acquire) >= want; }, "exchange_uv")) return 1;
            memcpy(m->slot(p, o), m->uvhs[d], ns[d] * sizeof(float));
            b->seq.store(m->msg_no, std::memory_order_release);
        }
        for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {
            ShmBox *b = m->box(m->rank, d);
            if (spin_until([&] { return b->seq.load(std::memory_order_acquire) >= m->msg_no; }, "exchange_uv")) return 1;
            memcpy(m->uvhr[d], m->slot(m->rank, d), nr[d] * sizeof(float));
            b->ack.store(m->msg_no, std::memory_order_release);
            HIPCHK(hipMemcpyAsync(m->uvr[d], m->uvhr[d], nr[d] * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
    }
    for (int d = 0; d < 4; ++d) if (has_peer(m, d)) {       // north, south, then east, west
        size_t off = 0;
        for (int b = 0; b < 2; ++b) {
            const UvBox &x = rb[d][b];
            if (icar_box_copy(c, x.f, which, x.i0, x.ni, x.j0, x.nj, m->uvr[d] + off, true)) return 1;
            off += (size_t)x.ni * x.nj * dd.nz;
        }
    }
    return 0;
}

// ---- co_min (time_step.f90:413) and the device-side maximum used by update_dt ----------------------------------------
static int host_reduce(IcarComm *m, double *v, bool take_min)
{
    ++m->red_no;
    ShmRed *mine = m->red(m->rank);
    mine->val[m->red_no & 1] = *v;
    mine->seq.store(m->red_no, std::memory_order_release);
    double r = *v;
    for (int p = 0; p < m->nranks; ++p) {
        ShmRed *o = m->red(p);
        if (spin_until([&] { return o->seq.load(std::memory_order_acquire) >= m->red_no; }, "co_min")) return 1;
        const double x = o->val[m->red_no & 1];
        r = take_min ? (x < r ? x : r) : (x > r ? x : r);
    }
    *v = r;
    return 0;
}

int icar_comm_co_reduce(icar_hip_ctx *c, double *value, bool take_min)
{
    IcarComm *m = c->comm;
    if (!m || (m->nranks == 1 && m->kind != ICAR_COMM_RCCL)) return 0;
    if (m->kind == ICAR_COMM_HOST) return host_reduce(m, value, take_min);
    if (m->kind != ICAR_COMM_RCCL) { icar_set_error("co_min: several images but no transport"); return 1; }
    *m->h_red = *value;
    HIPCHK(hipMemcpyAsync(m->d_red, m->h_red, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCHK(g_rccl.AllReduce(m->d_red, m->d_red, 1, ncclDouble, take_min ? ncclMin : ncclMax, m->nccl, c->stream));
    HIPCHK(hipMemcpyAsync(m->h_red, m->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *value = *m->h_red;
    return 0;
}

// max over images of one REAL(4) that is already in device memory (the CFL reduction's output): all-reduce in place on the
// context's stream, nothing waits.  Only the RCCL transport keeps the value on the device; returns 2 if the caller has to go
// through the host (icar_comm_co_reduce).
int icar_comm_max_device(icar_hip_ctx *c, float *d_val)
{
    IcarComm *m = c->comm;
    if (!m || m->kind != ICAR_COMM_RCCL) return 2;
    NCHK(g_rccl.AllReduce(d_val, d_val, 1, ncclFloat, ncclMax, m->nccl, c->stream));
    return 0;
}

void icar_comm_free(icar_hip_ctx *c)
{
    IcarComm *m = c->comm;
    if (!m) return;
    for (int d = 0; d < 4; ++d) {
        if (m->sbuf[d]) hipFree(m->sbuf[d]);
        if (m->rbuf[d]) hipFree(m->rbuf[d]);
        if (m->hs[d]) hipHostFree(m->hs[d]);
        if (m->hr[d]) hipHostFree(m->hr[d]);
        if (m->uvs[d]) hipFree(m->uvs[d]);
        if (m->uvr[d]) hipFree(m->uvr[d]);
        if (m->uvhs[d]) hipHostFree(m->uvhs[d]);
        if (m->uvhr[d]) hipHostFree(m->uvhr[d]);
    }
    if (m->d_red) hipFree(m->d_red);
    if (m->h_red) hipHostFree(m->h_red);
    if (m->nccl) g_rccl.CommDestroy(m->nccl);
    if (m->shm) {
        munmap(m->shm, m->shm_bytes);
        if (m->rank == 0) shm_unlink(m->shm_name.c_str());
    }
    delete m;
    c->comm = nullptr;
}

static int check_neighbors(int nranks, const int nb[4])
{
    for (int d = 0; d < 4; ++d)
        if (nb[d] >= nranks || nb[d] < ICAR_NEIGHBOR_SELF) { icar_set_error("comm_init: neighbors[] holds a rank of the communicator, ICAR_NEIGHBOR_NONE or ICAR_NEIGHBOR_SELF"); return 1; }
    return 0;
}

extern "C" {

int icar_hip_comm_unique_id(char uid[128])
{
    if (!uid) { icar_set_error("comm_unique_id: null argument"); return 1; }
    if (rccl_load()) return 1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id));
    memcpy(uid, &id, sizeof id);
    return 0;
}

int icar_hip_comm_init(icar_hip_ctx *c, int nranks, int rank, const char uid[128], const int neighbors[4])
{
    if (!c || !neighbors) { icar_set_error("comm_init: null argument"); return 1; }
    if (nranks < 1 || rank < 0 || rank >= nranks) { icar_set_error("comm_init: bad rank"); return 1; }
    if (check_neighbors(nranks, neighbors)) return 1;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    icar_comm_free(c);
    IcarComm *m = new IcarComm();
    m->nranks = nranks; m->rank = rank;
    memcpy(m->nb, neighbors, sizeof m->nb);
    c->comm = m;
    if (!uid) {
        bool peers = false;
        for (int d = 0; d < 4; ++d) peers = peers || neighbors[d] >= 0;
        if (nranks != 1 || peers) { icar_comm_free(c); icar_set_error("comm_init: several images need the unique id of icar_hip_comm_unique_id (broadcast from image 1)"); return 1; }
        m->kind = ICAR_COMM_LOCAL;
        return 0;
    }
    m->kind = ICAR_COMM_RCCL;
    if (rccl_load()) { icar_comm_free(c); return 1; }
    ncclUniqueId id; memcpy(&id, uid, sizeof id);
    if (rccl_check(g_rccl.CommInitRank(&m->nccl, nranks, id, rank), "ncclCommInitRank")) { m->nccl = nullptr; icar_comm_free(c); return 1; }
    if (icar_hip_check(hipMalloc(&m->d_red, 16), "hipMalloc") || icar_hip_check(hipHostMalloc((void **)&m->h_red, 16, hipHostMallocDefault), "hipHostMalloc")) { icar_comm_free(c); return 1; }
    return 0;
}

int icar_hip_comm_init_host(icar_hip_ctx *c, int nranks, int rank, const char *shm_name, size_t slot_bytes, const int neighbors[4])
{
    if (!c || !neighbors || !shm_name) { icar_set_error("comm_init_host: null argument"); return 1; }
    if (nranks < 1 || rank < 0 || rank >= nranks) { icar_set_error("comm_init_host: bad rank"); return 1; }
    if (check_neighbors(nranks, neighbors)) return 1;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    icar_comm_free(c);
    IcarComm *m = new IcarComm();
    m->kind = ICAR_COMM_HOST; m->nranks = nranks; m->rank = rank;
    memcpy(m->nb, neighbors, sizeof m->nb);
    m->shm_name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
    m->slot_bytes = (slot_bytes + 63) & ~(size_t)63;
    m->shm_bytes = 64 + (size_t)nranks * 5 * 64 + (size_t)nranks * 4 * m->slot_bytes;
    c->comm = m;
    // every image opens (creating if need be) and sizes the same object; ftruncate to an equal size is idempotent and a new
    // object reads as zeros, so there is no creation order to respect.  The caller picks a name that is unique to the run.
    const int fd = shm_open(m->shm_name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)m->shm_bytes) != 0) { if (fd >= 0) close(fd); icar_comm_free(c); icar_set_error("comm_init_host: shm_open / ftruncate failed"); return 1; }
    void *p = mmap(nullptr, m->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { icar_comm_free(c); icar_set_error("comm_init_host: mmap failed"); return 1; }
    m->shm = p;
    ShmHeader *hd = (ShmHeader *)p;
    if (rank == 0) { hd->nranks = (uint64_t)nranks; hd->slot_bytes = m->slot_bytes; std::atomic_thread_fence(std::memory_order_release); hd->magic = kMagic; }
    if (spin_until([&] { return ((volatile ShmHeader *)hd)->magic == kMagic; }, "comm_init_host")) { icar_comm_free(c); return 1; }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (hd->nranks != (uint64_t)nranks || hd->slot_bytes != m->slot_bytes) { icar_comm_free(c); icar_set_error("comm_init_host: the images disagree about nranks / slot_bytes"); return 1; }
    // leave only when everybody has mapped the object (rank 0 unlinks it when it is destroyed)
    double one = 1.0;
    if (host_reduce(m, &one, true)) { icar_comm_free(c); return 1; }
    return 0;
}

int icar_hip_comm_destroy(icar_hip_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm && c->comm->kind == ICAR_COMM_HOST && c->comm->nranks > 1) { double one = 1.0; host_reduce(c->comm, &one, true); }   // nobody unmaps while a neighbour still reads
    icar_comm_free(c);
    return 0;
}

int icar_hip_halo_send(icar_hip_ctx *c, int halo, const int *fields, int nfields)
{
    if (!c || (nfields > 0 && !fields)) { icar_set_error("halo_send: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_comm_halo_send(c, halo, fields, nfields);
}

int icar_hip_halo_retrieve(icar_hip_ctx *c, int halo, const int *fields, int nfields)
{
    if (!c || (nfields > 0 && !fields)) { icar_set_error("halo_retrieve: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_comm_halo_retrieve(c, halo, fields, nfields);
}

int icar_hip_exchange_uv(icar_hip_ctx *c, int halo, int update)
{
    if (!c) { icar_set_error("exchange_uv: null ctx"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    const int r = icar_comm_exchange_uv(c, halo, update ? 1 : 0);
    if (!r && !update) icar_winds_changed(c);
    return r;
}

int icar_hip_co_min(icar_hip_ctx *c, double *value)
{
    if (!c || !value) { icar_set_error("co_min: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_comm_co_reduce(c, value, true);
}

int icar_hip_co_max(icar_hip_ctx *c, double *value)
{
    if (!c || !value) { icar_set_error("co_max: null argument"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    return icar_comm_co_reduce(c, value, false);
}

int icar_hip_comm_kind(icar_hip_ctx *c) { return (c && c->comm) ? c->comm->kind : ICAR_COMM_NONE; }

int icar_hip_comm_timeout(double seconds)
{
    if (!(seconds > 0.0)) { icar_set_error("comm_timeout: seconds > 0"); return 1; }
    g_wait_seconds = seconds;
    return 0;
}

// How many images the transport itself says it connects: ncclCommCount of the RCCL communicator, the header of the shared
// segment of the host-staged transport, 1 without a transport.
int icar_hip_comm_ranks(icar_hip_ctx *c, int *nranks)
{
    if (!c || !nranks) { icar_set_error("comm_ranks: null argument"); return 1; }
    *nranks = 1;
    IcarComm *m = c->comm;
    if (!m) return 0;
    if (m->kind == ICAR_COMM_RCCL) { int n = 0; NCHK(g_rccl.CommCount(m->nccl, &n)); *nranks = n; }
    else if (m->kind == ICAR_COMM_HOST) *nranks = (int)((ShmHeader *)m->shm)->nranks;
    return 0;
}

// One halo_send / halo_retrieve (exchangeable_obj.f90:138-356) of a field stamped with this image's rank + 1, checked on the
// device: every halo cell (corners aside: they ride on the N / S rows and carry the neighbour's own halo) must hold the stamp
// of the neighbour on that side.  The field's contents are put back.  n_bad = cells that do not.
int icar_hip_halo_selfcheck(icar_hip_ctx *c, int halo, int *n_bad)
{
    if (!c || !n_bad) { icar_set_error("halo_selfcheck: null argument"); return 1; }
    IcarComm *m = c->comm;
    if (!m) { icar_set_error("halo_selfcheck: no transport (icar_hip_comm_init)"); return 1; }
    HIPCHK(hipSetDevice(c->device));
    const int fid = ICAR_F_WATER_VAPOR;
    float *f = icar_field_f(c, fid);
    if (!f) return 1;
    float *keep = nullptr; int *d_bad = nullptr;
    HIPCHK(hipMalloc(&keep, c->n3 * sizeof(float)));
    if (icar_hip_check(hipMalloc(&d_bad, sizeof(int)), "hipMalloc")) { (void)hipFree(keep); return 1; }
    int rc = 1;
    do {
        if (icar_hip_check(hipMemcpyAsync(keep, f, c->n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "save")) break;
        if (icar_hip_check(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream), "memset")) break;
        hipLaunchKernelGGL(k_stamp_fill, dim3((unsigned)((c->n3 + 255) / 256)), dim3(256), 0, c->stream, f, c->n3, (float)(m->rank + 1));
        if (icar_comm_halo_send(c, halo, &fid, 1)) break;
        if (icar_comm_halo_retrieve(c, halo, &fid, 1)) break;
        float want[4];
        for (int d = 0; d < 4; ++d) want[d] = m->nb[d] >= 0 ? (float)(m->nb[d] + 1) : m->nb[d] == ICAR_NEIGHBOR_SELF ? (float)(m->rank + 1) : -1.0f;
        hipLaunchKernelGGL(k_stamp_check, dim3((unsigned)((c->n3 + 255) / 256)), dim3(256), 0, c->stream, c->d, f, halo, want[0], want[1], want[2], want[3], d_bad);
        if (icar_hip_check(hipGetLastError(), "stamp check")) break;
        if (icar_hip_check(hipMemcpyAsync(f, keep, c->n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "restore")) break;
        if (icar_hip_check(hipMemcpyAsync(n_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream), "read")) break;
        if (icar_hip_check(hipStreamSynchronize(c->stream), "sync")) break;
        rc = 0;
    } while (0);
    (void)hipFree(keep); (void)hipFree(d_bad);
    return rc;
}

}  // extern "C"
