// somar_amd/csrc/comm_shm.cpp -- host-staged single-node transport over POSIX shared memory.
//
// Purpose: rehearse the sharded (one process per rank) data path -- packed halo messages, inter-level
// copiers, register exchanges, scalar reductions -- where RCCL cannot be used, e.g. several ranks sharing ONE
// GPU on a development box (RCCL refuses two ranks on one device).  Same Comm interface as comm_rccl.cpp, same
// message plans; only the wire differs: device -> shared host segment -> device, with pairwise sequence-number
// handshakes (neighbour exchange) or a barrier (allreduce) in between.  Synchronous and slow by design;
// production runs use RCCL over xGMI.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>

#include "level.h"

namespace somar {

namespace {
struct ShmHeader {
    std::atomic<int> arrive;
    std::atomic<int> generation;
    int nranks;
    int pad_;
    // directory: where in src's outbox the message for dst starts, and its length (doubles)
    long long off[16][16];
    long long cnt[16][16];
    std::atomic<long long> seq[16][16];  // seq[src][dst]: messages src has published for dst
    std::atomic<long long> ack[16][16];  // ack[src][dst]: messages of src that dst has consumed
    double red[16][64];  // allreduce slots
};
}  // namespace

struct ShmComm : Comm {
    std::string name;
    size_t box_bytes = 0;
    ShmHeader* hdr = nullptr;
    char* base = nullptr;
    size_t total = 0;
    int fd = -1;
    bool creator = false;

    double* outbox(int r) const { return reinterpret_cast<double*>(base + sizeof(ShmHeader) + (size_t)r * box_bytes); }

    void barrier()
    {
        const int gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrive.fetch_add(1, std::memory_order_acq_rel) == size - 1) {
            hdr->arrive.store(0, std::memory_order_relaxed);
            hdr->generation.store(gen + 1, std::memory_order_release);
        } else {
            while (hdr->generation.load(std::memory_order_acquire) == gen) sched_yield();
        }
    }

    ~ShmComm() override
    {
        if (base) munmap(base, total);
        if (fd >= 0) close(fd);
        if (creator) shm_unlink(name.c_str());
    }

    void allreduce(double* dbuf, int n, int op, hipStream_t st) override
    {
        if (size == 1) return;
        SOMAR_CHECK(n <= 64, "shm allreduce: too many values");
        // every copy goes through the caller's stream: a plain hipMemcpy runs on the null stream, which the
        // solver's non-blocking stream does not wait for (and an H2D copy from pageable memory may return before
        // its DMA has landed)
        SOMAR_HIP(hipMemcpyAsync(hdr->red[rank], dbuf, n * sizeof(double), hipMemcpyDeviceToHost, st));
        SOMAR_HIP(hipStreamSynchronize(st));
        barrier();
        double acc[64];
        for (int i = 0; i < n; ++i) acc[i] = hdr->red[0][i];
        for (int r = 1; r < size; ++r)  // rank order: every rank gets the same bits
            for (int i = 0; i < n; ++i) acc[i] = op ? (acc[i] > hdr->red[r][i] ? acc[i] : hdr->red[r][i]) : acc[i] + hdr->red[r][i];
        SOMAR_HIP(hipMemcpyAsync(dbuf, acc, n * sizeof(double), hipMemcpyHostToDevice, st));
        SOMAR_HIP(hipStreamSynchronize(st));
        barrier();
    }

    long long sent[16] = {0}, got[16] = {0};
    std::vector<int> prev_peers;

    void neighbor_exchange(const double* sendbuf, double* recvbuf, const std::vector<int>& peers,
                           const std::vector<long long>& soff, const std::vector<long long>& scount,
                           const std::vector<long long>& roff, const std::vector<long long>& rcount,
                           hipStream_t st) override
    {
        // pairwise handshakes (like grouped send/recv, only the ranks in `peers` take part): publish my outbox
        // with a per-pair sequence number, consume each peer's, acknowledge.  Peer lists are symmetric: if q is
        // in my list I am in q's.
        SOMAR_HIP(hipStreamSynchronize(st));
        for (int q : prev_peers)  // my previous message must have been consumed before the outbox is reused
            while (hdr->ack[rank][q].load(std::memory_order_acquire) < sent[q]) sched_yield();
        long long tot = 0;
        for (size_t q = 0; q < peers.size(); ++q) tot = std::max(tot, soff[q] + scount[q]);
        SOMAR_CHECK((size_t)tot * sizeof(double) <= box_bytes, "shm outbox too small for this message");
        if (tot) {
            SOMAR_HIP(hipMemcpyAsync(outbox(rank), sendbuf, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, st));
            SOMAR_HIP(hipStreamSynchronize(st));
        }
        for (size_t q = 0; q < peers.size(); ++q) {
            hdr->off[rank][peers[q]] = soff[q];
            hdr->cnt[rank][peers[q]] = scount[q];
            hdr->seq[rank][peers[q]].store(++sent[peers[q]], std::memory_order_release);
        }
        for (size_t q = 0; q < peers.size(); ++q) {
            const int r = peers[q];
            ++got[r];
            while (hdr->seq[r][rank].load(std::memory_order_acquire) < got[r]) sched_yield();
            SOMAR_CHECK(hdr->cnt[r][rank] == rcount[q], "shm exchange: send/receive counts disagree");
            if (rcount[q]) {
                SOMAR_HIP(hipMemcpyAsync(recvbuf + roff[q], outbox(r) + hdr->off[r][rank],
                                         (size_t)rcount[q] * sizeof(double), hipMemcpyHostToDevice, st));
                SOMAR_HIP(hipStreamSynchronize(st));  // the peer may reuse its outbox once acknowledged
            }
            hdr->ack[r][rank].store(got[r], std::memory_order_release);
        }
        prev_peers = peers;
    }
};

Comm* shm_create(const char* name, int rank, int nranks, size_t outbox_bytes)
{
    SOMAR_CHECK(nranks >= 1 && nranks <= 16 && rank >= 0 && rank < nranks, "shm comm: 1..16 ranks");
    ShmComm* c = new ShmComm;
    c->rank = rank;
    c->size = nranks;
    c->name = name;
    c->box_bytes = outbox_bytes;
    c->total = sizeof(ShmHeader) + (size_t)nranks * outbox_bytes;
    try {
        if (rank == 0) {
            shm_unlink(name);
            c->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            SOMAR_CHECK(c->fd >= 0, "shm_open(create) failed");
            c->creator = true;
            SOMAR_CHECK(ftruncate(c->fd, (off_t)c->total) == 0, "ftruncate failed");
        } else {
            for (int tries = 0; tries < 20000 && c->fd < 0; ++tries) {
                c->fd = shm_open(name, O_RDWR, 0600);
                if (c->fd < 0) usleep(1000);
            }
            SOMAR_CHECK(c->fd >= 0, "shm_open(attach) timed out");
            // wait until rank 0 has sized the segment
            for (int tries = 0; tries < 20000; ++tries) {
                off_t sz = lseek(c->fd, 0, SEEK_END);
                if (sz >= (off_t)c->total) break;
                usleep(1000);
            }
        }
        void* m = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
        SOMAR_CHECK(m != MAP_FAILED, "mmap failed");
        c->base = static_cast<char*>(m);
        c->hdr = reinterpret_cast<ShmHeader*>(m);
        if (rank == 0) {
            std::memset(static_cast<void*>(c->hdr), 0, sizeof(ShmHeader));
            c->hdr->nranks = nranks;
            std::atomic_thread_fence(std::memory_order_release);
            c->hdr->generation.store(1, std::memory_order_release);  // "initialised"
        } else {
            while (c->hdr->generation.load(std::memory_order_acquire) == 0) usleep(1000);
        }
        c->barrier();
    } catch (...) {
        delete c;
        throw;
    }
    return c;
}

}  // namespace somar
