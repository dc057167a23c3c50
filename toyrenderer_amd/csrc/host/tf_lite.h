// tf_lite.h -- the slice of Taskflow the frame schedule uses (extern/taskflow is an empty
// submodule in the reference snapshot): Taskflow::emplace / placeholder, Task::succeed / precede,
// Executor::corun.  Tasks run on a small persistent pool of std::threads in dependency order, so Render()
// really is called from arbitrary worker threads like in the reference (RenderGraph.cpp:254-288).
#pragma once

#include <atomic>
#include <condition_variable>
#include <exception>
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
    // The workers are created once and parked on a condition variable between frames (spawning
    // threads per corun costs ~0.1 ms per frame, as much as recording the whole visibility path).
    explicit Executor(unsigned workers = 4)
    {
        for (unsigned i = 1; i < (workers ? workers : 1u); ++i) m_Threads.emplace_back([this] { workerLoop(); });
    }
    ~Executor()
    {
        {
            std::lock_guard<std::mutex> lk(m_Mutex);
            m_Stop = true;
        }
        m_Cv.notify_all();
        for (std::thread& t : m_Threads) t.join();
    }
    Executor(const Executor&) = delete;
    Executor& operator=(const Executor&) = delete;

    // Runs every task of the flow (the caller participates), returns when all are done.
    void corun(Taskflow& flow)
    {
        const size_t n = flow.m_Nodes.size();
        if (n == 0) return;
        Job job;
        job.flow = &flow;
        job.pending.resize(n);
        for (size_t i = 0; i < n; ++i) {
            job.pending[i] = flow.m_Nodes[i].numDeps;
            if (flow.m_Nodes[i].numDeps == 0) job.ready.push_back(i);
        }
        std::unique_lock<std::mutex> lk(m_Mutex);
        m_Job = &job;
        if (job.ready.size() > 1) m_Cv.notify_all();
        for (;;) {
            runReady(lk);
            if (job.done == n) break;
            m_Cv.wait(lk, [&] { return job.done == n || !job.ready.empty(); });
        }
        m_Job = nullptr;                       // no task is running any more: nobody else refers to `job`
        lk.unlock();
        if (job.error) std::rethrow_exception(job.error);   // first exception a task threw, on whichever thread it ran
    }

private:
    struct Job
    {
        Taskflow* flow = nullptr;
        std::vector<size_t> pending, ready;
        size_t done = 0;
        std::exception_ptr error;
    };

    // Called with the lock held; returns with it held.
    void runReady(std::unique_lock<std::mutex>& lk)
    {
        while (m_Job && !m_Job->ready.empty()) {
            Job* job = m_Job;
            const size_t idx = job->ready.back();
            job->ready.pop_back();
            lk.unlock();
            std::exception_ptr thrown;
            try { job->flow->m_Nodes[idx].fn(); } catch (...) { thrown = std::current_exception(); }
            lk.lock();
            if (thrown && !job->error) job->error = thrown;
            ++job->done;
            size_t released = 0;
            for (size_t s : job->flow->m_Nodes[idx].successors)
                if (--job->pending[s] == 0) { job->ready.push_back(s); ++released; }
            if (released > 1 || job->done == job->flow->m_Nodes.size()) m_Cv.notify_all();
        }
    }

    void workerLoop()
    {
        std::unique_lock<std::mutex> lk(m_Mutex);
        for (;;) {
            m_Cv.wait(lk, [&] { return m_Stop || (m_Job && !m_Job->ready.empty()); });
            if (m_Stop) return;
            runReady(lk);
        }
    }

    std::mutex m_Mutex;
    std::condition_variable m_Cv;
    Job* m_Job = nullptr;
    bool m_Stop = false;
    std::vector<std::thread> m_Threads;
};

} // namespace tf
