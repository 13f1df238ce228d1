// One gzip member written by several threads.
//
// The reference pipes its .zdw stream into a `gzip` child (ConvertToZDW.cpp:817-835); one core deflates about 30 MB of ZDW per
// second, which is ~100 MB/s of input text while the device encodes hundreds of GB/s.  Here the stream is cut into pieces of
// 2 MiB; each piece is deflated on its own (raw deflate, ended by an empty stored block on a byte boundary: Z_SYNC_FLUSH), the
// pieces are written in order behind one gzip header and followed by one final block and the trailer (CRC-32 of everything, put
// together from the pieces' with crc32_combine; length mod 2^32).  What comes out is an ordinary single-member .gz: gzip -d, zlib's
// gz* functions and java.util.zip.GZIPInputStream read it like the child's output.  The compressed BYTES differ from the child's
// (as they do between gzip versions); the .zdw stream inside is the same.
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace zdwhost {

class ParallelGzip {
public:
    static constexpr size_t PIECE = 2u << 20;

    // f: opened for writing, owned by the caller until close().  threads >= 1; level 1..9.
    ParallelGzip(FILE *f, unsigned threads, int level) : f_(f), level_(level)
    {
        static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};      // deflate, no name, no time, Unix
        if (fwrite(hdr, 1, sizeof hdr, f_) != sizeof hdr) failed_ = true;
        cur_.reserve(PIECE);
        max_pending_ = 3 * (threads ? threads : 1) + 2;
        for (unsigned i = 0; i < (threads ? threads : 1); ++i) workers_.emplace_back([this] { work(); });
    }
    ParallelGzip(const ParallelGzip &) = delete;
    ParallelGzip &operator=(const ParallelGzip &) = delete;
    ~ParallelGzip() { stop(); }

    // (one caller at a time)
    bool write(const void *p, size_t n)
    {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        while (n) {
            const size_t k = n < PIECE - cur_.size() ? n : PIECE - cur_.size();
            cur_.insert(cur_.end(), b, b + k);
            b += k; n -= k;
            if (cur_.size() == PIECE && !submit()) return false;
        }
        return !failed_;
    }

    // the rest of the stream, the final block and the trailer; the file is flushed, not closed.  False: a write or a deflate failed
    bool close()
    {
        if (closed_) return !failed_;
        closed_ = true;
        if (!cur_.empty()) (void)submit();
        drain(0);
        stop();
        static const unsigned char last[2] = {0x03, 0x00};                               // an empty fixed block, marked final
        unsigned char tr[8];
        for (int i = 0; i < 4; ++i) { tr[i] = (unsigned char)(crc_ >> (8 * i)); tr[4 + i] = (unsigned char)(total_ >> (8 * i)); }
        if (!failed_ && (fwrite(last, 1, 2, f_) != 2 || fwrite(tr, 1, 8, f_) != 8 || fflush(f_) != 0)) failed_ = true;
        return !failed_;
    }

private:
    struct Job {
        std::vector<unsigned char> in, out;
        uLong crc = 0;
        bool done = false, bad = false;
    };

    bool submit()
    {
        auto j = std::make_shared<Job>();
        j->in.swap(cur_);
        cur_.reserve(PIECE);
        {
            std::lock_guard<std::mutex> g(m_);
            todo_.push_back(j);
            order_.push_back(j);
        }
        have_work_.notify_one();
        drain(max_pending_);
        return !failed_;
    }

    // writes finished pieces in order; waits while more than `keep` are outstanding
    void drain(size_t keep)
    {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (order_.empty()) return;
                if (!order_.front()->done) {
                    if (order_.size() <= keep) return;
                    piece_done_.wait(lk, [&] { return order_.front()->done; });
                }
                j = order_.front();
                order_.pop_front();
            }
            if (j->bad) failed_ = true;
            if (!failed_ && fwrite(j->out.data(), 1, j->out.size(), f_) != j->out.size()) failed_ = true;
            crc_ = total_ ? crc32_combine(crc_, j->crc, (z_off_t)j->in.size()) : j->crc;
            total_ += j->in.size();
        }
    }

    void work()
    {
        z_stream z;
        memset(&z, 0, sizeof z);
        const bool ready = deflateInit2(&z, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                have_work_.wait(lk, [&] { return stopping_ || !todo_.empty(); });
                if (todo_.empty()) break;
                j = todo_.front();
                todo_.pop_front();
            }
            bool bad = !ready;
            if (ready) {
                (void)deflateReset(&z);
                j->out.resize(deflateBound(&z, (uLong)j->in.size()) + 16);
                z.next_in = j->in.data(); z.avail_in = (uInt)j->in.size();
                z.next_out = j->out.data(); z.avail_out = (uInt)j->out.size();
                const int rc = deflate(&z, Z_SYNC_FLUSH);
                bad = rc != Z_OK || z.avail_in != 0 || z.avail_out == 0;
                j->out.resize(j->out.size() - z.avail_out);
                j->crc = crc32(crc32(0L, Z_NULL, 0), j->in.data(), (uInt)j->in.size());
            }
            {
                std::lock_guard<std::mutex> g(m_);
                j->bad = bad;
                j->done = true;
            }
            piece_done_.notify_all();
        }
        if (ready) (void)deflateEnd(&z);
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stopping_ = true;
        }
        have_work_.notify_all();
        for (auto &t : workers_) if (t.joinable()) t.join();
        workers_.clear();
    }

    FILE *f_;
    int level_;
    bool failed_ = false, closed_ = false, stopping_ = false;
    size_t max_pending_ = 8;
    std::vector<unsigned char> cur_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable have_work_, piece_done_;
    std::deque<std::shared_ptr<Job> > todo_, order_;
    uLong crc_ = 0;
    unsigned long long total_ = 0;
};

}  // namespace zdwhost
