// stable_lock_file.cpp — see stable_lock_file.hpp.
#include "stable_lock_file.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <vector>

#include "wire.hpp"

namespace raftgpu {
namespace host {

namespace {
int64_t be64(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return (int64_t)v; }
int32_t be32(const unsigned char *p) { uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[i]; return (int32_t)v; }
void put64(std::string &o, int64_t x) { for (int s = 56; s >= 0; s -= 8) o.push_back((char)((uint64_t)x >> s)); }
void put32(std::string &o, int32_t x) { for (int s = 24; s >= 0; s -= 8) o.push_back((char)((uint32_t)x >> s)); }
bool fail(std::string *err, const std::string &what) { if (err) *err = what; return false; }
}  // namespace

bool read_stable_lock(const std::string &path, StableLockImage *out, std::string *err)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(err, "open " + path + ": " + strerror(errno));
    std::vector<unsigned char> buf(1 << 16);
    size_t n = 0;
    for (;;) {
        const ssize_t r = ::read(fd, buf.data() + n, buf.size() - n);
        if (r < 0) { if (errno == EINTR) continue; const std::string e = strerror(errno); ::close(fd); return fail(err, "read " + path + ": " + e); }
        if (r == 0) break;
        n += (size_t)r;
        if (n == buf.size()) { ::close(fd); return fail(err, path + ": larger than any StableLock file"); }
    }
    ::close(fd);
    if (n < 28) return fail(err, path + ": shorter than the 28-byte header");
    StableLockImage img;
    img.milestone_index = be64(buf.data()); img.milestone_term = be64(buf.data() + 8); img.term = be64(buf.data() + 16);
    const int32_t len = be32(buf.data() + 24);
    if (len < 0 || (size_t)len > n - 28) return fail(err, path + ": the id's length field points beyond the file");
    if (len > 0) {                                       // StableLock.restore: `if (length > 0) candidate = Serialization.readObject(bytes)`
        rafting::wire::KryoBodyCodec::Node node;
        bool is_null = false;
        if (!rafting::wire::KryoBodyCodec::decode_node_object(reinterpret_cast<const char *>(buf.data() + 28), (size_t)len, node, is_null))
            return fail(err, path + ": the id bytes are not Kryo's image of a NodeID");
        if (!is_null) { img.has_vote = true; img.vote_host = node.hostname; img.vote_port = node.port; }
    }
    *out = img;
    return true;
}

bool write_stable_lock(const std::string &path, const StableLockImage &img, std::string *err)
{
    std::string id;
    if (img.has_vote) {
        const rafting::wire::KryoBodyCodec::Node node{img.vote_host, img.vote_port};
        rafting::wire::KryoBodyCodec::encode_node_object(&node, id);
    } else {
        rafting::wire::KryoBodyCodec::encode_node_object(nullptr, id);     // persist(term, null) writes Kryo's null: one byte
    }
    std::string bytes;
    put64(bytes, img.milestone_index); put64(bytes, img.milestone_term); put64(bytes, img.term); put32(bytes, (int32_t)id.size());
    bytes += id;
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT, 0644);      // (no O_TRUNC: the reference does not truncate either)
    if (fd < 0) return fail(err, "open " + path + ": " + strerror(errno));
    size_t off = 0;
    while (off < bytes.size()) {
        const ssize_t w = ::pwrite(fd, bytes.data() + off, bytes.size() - off, (off_t)off);
        if (w < 0) { if (errno == EINTR) continue; const std::string e = strerror(errno); ::close(fd); return fail(err, "write " + path + ": " + e); }
        off += (size_t)w;
    }
    const int rc = ::fsync(fd);
    ::close(fd);
    return rc == 0 ? true : fail(err, "fsync " + path);
}

}  // namespace host
}  // namespace raftgpu
