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
#include <sys/stat.h>
#include <time.h>
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
    long long pad2_;
    // attach handshake: rank r > 0 writes a random token into hello[r]; only a LIVE rank 0 copies it to echo[r].  A segment of
    // the same name left behind by an earlier run (opened before this run's rank 0 unlinks it) never answers.
    std::atomic<long long> hello[16];
    std::atomic<long long> echo[16];
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

    static constexpr size_t RED_DOUBLES = 16384;  // per-rank slot of the vector all-reduce (ordered sums of small levels)
    double* outbox(int r) const { return reinterpret_cast<double*>(base + sizeof(ShmHeader) + (size_t)r * box_bytes); }
    double* redbox(int r) const
    {
        return reinterpret_cast<double*>(base + sizeof(ShmHeader) + (size_t)size * box_bytes) + (size_t)r * RED_DOUBLES;
    }

    static double now_s()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + 1e-9 * ts.tv_nsec;
    }

    // a rank that never arrives (crashed peer, stale segment) must end in an error, not in a hang
    void barrier()
    {
        const int gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrive.fetch_add(1, std::memory_order_acq_rel) == size - 1) {
            hdr->arrive.store(0, std::memory_order_relaxed);
            hdr->generation.store(gen + 1, std::memory_order_release);
        } else {
            const double t0 = now_s();
            long spins = 0;
            while (hdr->generation.load(std::memory_order_acquire) == gen) {
                sched_yield();
                if ((++spins & 0xfff) == 0) SOMAR_CHECK(now_s() - t0 < 300.0, "shm barrier timed out (a peer rank is gone?)");
            }
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
        if (n > 64) {
            // vector form: each rank publishes its vector in its slot, every rank adds the slots in rank order
            SOMAR_CHECK((size_t)n <= RED_DOUBLES, "shm allreduce: too many values");
            SOMAR_HIP(hipMemcpyAsync(redbox(rank), dbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
            SOMAR_HIP(hipStreamSynchronize(st));
            barrier();
            std::vector<double> acc(redbox(0), redbox(0) + n);
            for (int r = 1; r < size; ++r) {
                const double* v = redbox(r);
                for (int i = 0; i < n; ++i) acc[i] = op ? (acc[i] > v[i] ? acc[i] : v[i]) : acc[i] + v[i];
            }
            SOMAR_HIP(hipMemcpyAsync(dbuf, acc.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            SOMAR_HIP(hipStreamSynchronize(st));
            barrier();
            return;
        }
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
    c->total = sizeof(ShmHeader) + (size_t)nranks * outbox_bytes + (size_t)nranks * ShmComm::RED_DOUBLES * sizeof(double);
    try {
        if (rank == 0) {
            shm_unlink(name);
            c->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            SOMAR_CHECK(c->fd >= 0, "shm_open(create) failed");
            c->creator = true;
            SOMAR_CHECK(ftruncate(c->fd, (off_t)c->total) == 0, "ftruncate failed");
            void* m = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
            SOMAR_CHECK(m != MAP_FAILED, "mmap failed");
            c->base = static_cast<char*>(m);
            c->hdr = reinterpret_cast<ShmHeader*>(m);
            std::memset(static_cast<void*>(c->hdr), 0, sizeof(ShmHeader));
            c->hdr->nranks = nranks;
            std::atomic_thread_fence(std::memory_order_release);
            c->hdr->generation.store(1, std::memory_order_release);  // "initialised"
            // answer every attaching rank's token (300 s for the slowest peer to get here, the barrier's own limit)
            const double t0 = ShmComm::now_s();
            int answered = 0;
            std::vector<char> done(nranks, 0);
            while (answered < nranks - 1) {
                SOMAR_CHECK(ShmComm::now_s() - t0 < 300.0, "shm create: not every rank attached within 300 s");
                for (int r = 1; r < nranks; ++r) {
                    if (done[r]) continue;
                    const long long t = c->hdr->hello[r].load(std::memory_order_acquire);
                    if (t != 0) {
                        c->hdr->echo[r].store(t, std::memory_order_release);
                        done[r] = 1;
                        ++answered;
                    }
                }
                usleep(200);
            }
        } else {
            // Attach: the segment must exist, have its full size, be initialised, be made for this many ranks and be
            // LIVE -- a segment of the same name left by an earlier run (opened before rank 0 unlinks it) is dead
            // memory on which the barrier below would never complete.  Liveness is a handshake, not a clock: this rank's
            // random token must come back from rank 0; while waiting, the name is re-opened now and then, and if it has
            // come to denote another segment (rank 0 unlinked the stale one and created its own) the attach starts over.
            const double t0 = ShmComm::now_s();
            bool ok = false;
            while (!ok) {
                SOMAR_CHECK(ShmComm::now_s() - t0 < 300.0, "shm attach timed out (no live, fully sized segment of that name appeared)");
                if (c->base) { munmap(c->base, c->total); c->base = nullptr; c->hdr = nullptr; }
                if (c->fd >= 0) { close(c->fd); c->fd = -1; }
                c->fd = shm_open(name, O_RDWR, 0600);
                if (c->fd < 0) { usleep(1000); continue; }
                if (lseek(c->fd, 0, SEEK_END) < (off_t)c->total) { usleep(1000); continue; }
                void* m = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
                SOMAR_CHECK(m != MAP_FAILED, "mmap failed");
                c->base = static_cast<char*>(m);
                c->hdr = reinterpret_cast<ShmHeader*>(m);
                if (c->hdr->generation.load(std::memory_order_acquire) == 0) { usleep(1000); continue; }
                if (c->hdr->nranks != nranks) { usleep(20000); continue; }
                timespec ts;
                clock_gettime(CLOCK_REALTIME, &ts);
                long long token = ((long long)getpid() << 32) ^ (long long)ts.tv_nsec ^ ((long long)ts.tv_sec << 20) ^ rank;
                if (token == 0) token = 1;
                c->hdr->echo[rank].store(0, std::memory_order_relaxed);
                c->hdr->hello[rank].store(token, std::memory_order_release);
                struct stat mine;
                SOMAR_CHECK(fstat(c->fd, &mine) == 0, "fstat failed");
                bool stale = false;
                long polls = 0;
                while (c->hdr->echo[rank].load(std::memory_order_acquire) != token) {
                    SOMAR_CHECK(ShmComm::now_s() - t0 < 300.0, "shm attach timed out (rank 0 never answered on a segment of that name)");
                    usleep(200);
                    if ((++polls % 100) == 0) {   // every ~20 ms: does the name still denote the segment I mapped?
                        const int fd2 = shm_open(name, O_RDONLY, 0600);
                        struct stat now;
                        const bool same = fd2 >= 0 && fstat(fd2, &now) == 0 && now.st_ino == mine.st_ino && now.st_dev == mine.st_dev;
                        if (fd2 >= 0) close(fd2);
                        if (!same) { stale = true; break; }
                    }
                }
                if (stale) continue;
                ok = true;
            }
        }
        c->barrier();
    } catch (...) {
        delete c;
        throw;
    }
    return c;
}

}  // namespace somar
