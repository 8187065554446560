// multi_device.cpp — see multi_device.hpp
#include "multi_device.hpp"

#include <stdexcept>

namespace raftgpu {
namespace host {

MultiDeviceManager::MultiDeviceManager(const std::vector<int> &devices, uint32_t maxContexts, uint32_t clusterSize, ID self, bool preVote)
    : capacity_(maxContexts)
{
    if (devices.empty()) throw std::invalid_argument("MultiDeviceManager needs at least one device");
    per_shard_ = (maxContexts + (uint32_t)devices.size() - 1) / (uint32_t)devices.size();      // ceil(G / N): block partition
    for (int dev : devices) {
        std::unique_ptr<Shard> s(new Shard());
        s->mgr.reset(new ContextManager(dev, per_shard_, clusterSize, self, preVote));
        Shard *raw = s.get();
        s->feeder = std::thread([this, raw] { feed(*raw); });
        shards_.push_back(std::move(s));
    }
}

MultiDeviceManager::~MultiDeviceManager()
{
    for (auto &s : shards_) {
        { std::lock_guard<std::mutex> lk(s->m); s->quit = true; }
        s->cv.notify_all();
        s->feeder.join();
    }
}

// the feeder thread of one device: waits for a drain request, runs the shard's flush (rg_submit + effects), reports back
void MultiDeviceManager::feed(Shard &s)
{
    std::unique_lock<std::mutex> lk(s.m);
    for (;;) {
        s.cv.wait(lk, [&] { return s.go || s.quit; });
        if (s.quit) return;
        s.go = false;
        const int64_t now = s.now;
        lk.unlock();
        std::vector<Outcome> out;
        std::exception_ptr err;
        try { std::lock_guard<std::mutex> wk(s.work); out = s.mgr->flush(now); } catch (...) { err = std::current_exception(); }
        lk.lock();
        s.out = std::move(out);
        s.error = err;
        s.done = true;
        s.cv.notify_all();
    }
}

RaftContext &MultiDeviceManager::createContext(const std::string &id, int64_t restoreTerm, ID restoreBallot)
{
    size_t k;
    {
        std::lock_guard<std::mutex> lk(route_m_);
        auto it = where_.find(id);
        if (it == where_.end()) {
            if (created_ >= capacity_) throw std::length_error("context table is full");
            const uint32_t gid = created_++;
            it = where_.emplace(id, std::make_pair((size_t)(gid / per_shard_), gid)).first;    // gpu = gid / ceil(G / N)
        }
        k = it->second.first;
    }
    // the shard's table is one C-ABI handle: wait for a drain of it that is running (ContextManager::createContext returns the existing
    // context when another thread won the race for this id)
    std::lock_guard<std::mutex> wk(shards_[k]->work);
    return shards_[k]->mgr->createContext(id, restoreTerm, restoreBallot);
}

RaftContext *MultiDeviceManager::getContext(const std::string &id)
{
    size_t k;
    {
        std::lock_guard<std::mutex> lk(route_m_);
        auto it = where_.find(id);
        if (it == where_.end()) return nullptr;
        k = it->second.first;
    }
    std::lock_guard<std::mutex> wk(shards_[k]->work);
    return shards_[k]->mgr->getContext(id);
}

size_t MultiDeviceManager::shardOf(const std::string &id) const { std::lock_guard<std::mutex> lk(route_m_); return where_.at(id).first; }
uint32_t MultiDeviceManager::globalGid(const std::string &id) const { std::lock_guard<std::mutex> lk(route_m_); return where_.at(id).second; }

std::vector<std::vector<Outcome>> MultiDeviceManager::flushAll(int64_t now)
{
    for (auto &s : shards_) {                                      // fan-out
        { std::lock_guard<std::mutex> lk(s->m); s->now = now; s->done = false; s->go = true; }
        s->cv.notify_all();
    }
    std::vector<std::vector<Outcome>> all(shards_.size());
    std::exception_ptr first;
    for (size_t k = 0; k < shards_.size(); k++) {                  // fan-in
        Shard &s = *shards_[k];
        std::unique_lock<std::mutex> lk(s.m);
        s.cv.wait(lk, [&] { return s.done; });
        all[k] = std::move(s.out);
        if (s.error && !first) first = s.error;
    }
    if (first) std::rethrow_exception(first);
    return all;
}

}  // namespace host
}  // namespace raftgpu
