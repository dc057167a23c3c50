// tf_lite.h -- the slice of Taskflow the frame schedule uses (extern/taskflow is an empty
// submodule in the reference snapshot): Taskflow::emplace / placeholder, Task::succeed / precede,
// Executor::corun.  Tasks run on a small pool of std::threads in dependency order, so Render()
// really is called from arbitrary worker threads like in the reference (RenderGraph.cpp:254-288).
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace tf
{

class Taskflow;

class Task
{
public:
    Task() = default;
    Task& succeed(const Task& other);
    Task& precede(const Task& other);
    bool empty() const { return m_Flow == nullptr; }

private:
    friend class Taskflow;
    friend class Executor;
    Task(Taskflow* f, size_t i) : m_Flow(f), m_Idx(i) {}
    Taskflow* m_Flow = nullptr;
    size_t m_Idx = 0;
};

class Taskflow
{
public:
    template <typename F> Task emplace(F&& f)
    {
        m_Nodes.push_back(Node{ std::function<void()>(std::forward<F>(f)), {}, 0 });
        return Task(this, m_Nodes.size() - 1);
    }
    Task placeholder() { return emplace([] {}); }
    void clear() { m_Nodes.clear(); }
    size_t num_tasks() const { return m_Nodes.size(); }

private:
    friend class Task;
    friend class Executor;
    struct Node { std::function<void()> fn; std::vector<size_t> successors; size_t numDeps; };
    std::vector<Node> m_Nodes;
};

inline Task& Task::succeed(const Task& other)
{
    m_Flow->m_Nodes[other.m_Idx].successors.push_back(m_Idx);
    m_Flow->m_Nodes[m_Idx].numDeps++;
    return *this;
}
inline Task& Task::precede(const Task& other)
{
    m_Flow->m_Nodes[m_Idx].successors.push_back(other.m_Idx);
    m_Flow->m_Nodes[other.m_Idx].numDeps++;
    return *this;
}

class Executor
{
public:
    explicit Executor(unsigned workers = 4) : m_Workers(workers ? workers : 1) {}

    // Runs every task of the flow (the caller participates), returns when all are done.
    void corun(Taskflow& flow)
    {
        const size_t n = flow.m_Nodes.size();
        if (n == 0) return;
        std::vector<std::atomic<size_t>> pending(n);
        std::vector<size_t> ready;
        for (size_t i = 0; i < n; ++i) {
            pending[i].store(flow.m_Nodes[i].numDeps);
            if (flow.m_Nodes[i].numDeps == 0) ready.push_back(i);
        }
        std::mutex mtx;
        std::condition_variable cv;
        size_t done = 0;
        auto worker = [&] {
            for (;;) {
                size_t idx;
                {
                    std::unique_lock<std::mutex> lk(mtx);
                    cv.wait(lk, [&] { return !ready.empty() || done == n; });
                    if (ready.empty()) return;
                    idx = ready.back();
                    ready.pop_back();
                }
                flow.m_Nodes[idx].fn();
                {
                    std::lock_guard<std::mutex> lk(mtx);
                    ++done;
                    for (size_t s : flow.m_Nodes[idx].successors)
                        if (pending[s].fetch_sub(1) == 1) ready.push_back(s);
                }
                cv.notify_all();
            }
        };
        std::vector<std::thread> threads;
        for (unsigned i = 1; i < m_Workers; ++i) threads.emplace_back(worker);
        worker();
        for (std::thread& t : threads) t.join();
    }

private:
    unsigned m_Workers;
};

} // namespace tf
