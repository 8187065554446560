// stable_store.cpp — see stable_store.hpp.
#include "stable_store.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <stdexcept>

namespace raftgpu {
namespace host {

namespace {
const char MAGIC[8] = {'R', 'G', 'S', 'S', '0', '0', '0', '1'};
const size_t REC = 24;

uint32_t crc32(const unsigned char *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return ~c;
}

void encode(unsigned char *out, const StableStore::Record &r, uint32_t seq)
{
    memcpy(out, &r.gid, 4); memcpy(out + 4, &r.votedFor, 4); memcpy(out + 8, &r.term, 8); memcpy(out + 16, &seq, 4);
    const uint32_t c = crc32(out, 20);
    memcpy(out + 20, &c, 4);
}

// A new or renamed directory entry is durable only once the DIRECTORY has been synced: fdatasync on the file does not cover it.
void sync_parent_dir(const std::string &path)
{
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : path.substr(0, slash));
    const int dfd = ::open(dir.c_str(), O_RDONLY | O_DIRECTORY);
    if (dfd < 0) throw std::runtime_error("StableStore open dir " + dir + ": " + strerror(errno));
    const int rc = ::fsync(dfd);
    const int err = errno;
    ::close(dfd);
    if (rc != 0) throw std::runtime_error("StableStore fsync dir " + dir + ": " + strerror(err));
}

void write_all(int fd, const unsigned char *p, size_t n)
{
    while (n) {
        ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("StableStore write: ") + strerror(errno)); }
        p += w; n -= (size_t)w;
    }
}
}  // namespace

StableStore::StableStore(const std::string &path) : path_(path)
{
    fd_ = ::open(path.c_str(), O_RDWR | O_CREAT, 0644);
    if (fd_ < 0) throw std::runtime_error("StableStore open " + path + ": " + strerror(errno));
    replay();
}

StableStore::~StableStore() { if (fd_ >= 0) ::close(fd_); }

void StableStore::replay()
{
    const off_t size = ::lseek(fd_, 0, SEEK_END);
    if (size == 0) {
        write_all(fd_, (const unsigned char *)MAGIC, sizeof MAGIC);
        if (::fdatasync(fd_) != 0) throw std::runtime_error("StableStore fdatasync");
        sync_parent_dir(path_);                             // the file's name must survive a crash too
        return;
    }
    std::vector<unsigned char> buf((size_t)size);
    if (::pread(fd_, buf.data(), buf.size(), 0) != (ssize_t)buf.size()) throw std::runtime_error("StableStore read");
    if (buf.size() < sizeof MAGIC || memcmp(buf.data(), MAGIC, sizeof MAGIC) != 0) throw std::runtime_error("StableStore: bad magic in " + path_);
    size_t off = sizeof MAGIC;
    for (; off + REC <= buf.size(); off += REC) {
        const unsigned char *p = buf.data() + off;
        uint32_t c;
        memcpy(&c, p + 20, 4);
        if (c != crc32(p, 20)) break;                       // torn tail: everything before it is intact
        Record r;
        memcpy(&r.gid, p, 4); memcpy(&r.votedFor, p + 4, 4); memcpy(&r.term, p + 8, 8); memcpy(&seq_, p + 16, 4);
        latest_[r.gid] = r;
        records_++;
    }
    if ((off_t)off != size && ::ftruncate(fd_, (off_t)off) != 0) throw std::runtime_error("StableStore ftruncate");
    ::lseek(fd_, (off_t)off, SEEK_SET);
}

void StableStore::persist(const std::vector<Record> &batch)
{
    if (batch.empty()) return;
    if (broken_) throw std::runtime_error("StableStore: the journal could not be cut back after a failed write; refusing further batches");
    std::vector<unsigned char> buf(batch.size() * REC);
    uint32_t seq = seq_;
    for (size_t i = 0; i < batch.size(); i++) encode(buf.data() + i * REC, batch[i], ++seq);
    const off_t start = ::lseek(fd_, 0, SEEK_CUR);
    try {
        write_all(fd_, buf.data(), buf.size());
        if (::fdatasync(fd_) != 0) throw std::runtime_error(std::string("StableStore fdatasync: ") + strerror(errno));
    } catch (...) {
        // nothing of a failed batch may be seen later: cut the file back to where the batch began (a partial record would
        // otherwise sit in FRONT of the next successful batch and make replay() stop there), and leave latest_/seq_ alone so
        // that restore() keeps answering with what IS durable
        if (start >= 0 && ::ftruncate(fd_, start) == 0) ::lseek(fd_, start, SEEK_SET);
        else broken_ = true;                                 // a torn record may sit in the file: fail-stop (replay() would cut there)
        throw;
    }
    seq_ = seq;                                              // the memory image moves only after the bytes are durable
    for (const Record &r : batch) latest_[r.gid] = r;
    syncs_++;
    records_ += batch.size();
}

bool StableStore::restore(uint32_t gid, int64_t *term, int32_t *votedFor) const
{
    auto it = latest_.find(gid);
    if (it == latest_.end()) return false;
    if (term) *term = it->second.term;
    if (votedFor) *votedFor = it->second.votedFor;
    return true;
}

void StableStore::compact()
{
    const std::string tmp = path_ + ".tmp";
    int fd = ::open(tmp.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw std::runtime_error("StableStore compact open: " + std::string(strerror(errno)));
    uint32_t seq = 0;
    try {
        std::vector<unsigned char> buf(sizeof MAGIC + latest_.size() * REC);
        memcpy(buf.data(), MAGIC, sizeof MAGIC);
        size_t i = 0;
        for (const auto &kv : latest_) encode(buf.data() + sizeof MAGIC + (i++) * REC, kv.second, ++seq);
        write_all(fd, buf.data(), buf.size());
        if (::fdatasync(fd) != 0) throw std::runtime_error(std::string("StableStore compact fdatasync: ") + strerror(errno));
        if (::rename(tmp.c_str(), path_.c_str()) != 0) throw std::runtime_error(std::string("StableStore compact rename: ") + strerror(errno));
    } catch (...) {
        ::close(fd);                                         // the live journal is untouched: the store keeps working on it
        ::unlink(tmp.c_str());
        throw;
    }
    ::close(fd_);                                            // the name now means the new file: append there from here on, whatever follows
    fd_ = fd;
    seq_ = seq;
    ::lseek(fd_, 0, SEEK_END);
    try {
        sync_parent_dir(path_);                              // make the rename itself durable before any later batch is acknowledged
    } catch (...) {
        // the directory entry of the new journal may not be durable: after a crash the OLD file could reappear under path_ without the
        // batches acknowledged from now on. Fail-stop like a failed persist(): nothing more is acknowledged through this object.
        broken_ = true;
        throw;
    }
    syncs_++;
}

}  // namespace host
}  // namespace raftgpu
