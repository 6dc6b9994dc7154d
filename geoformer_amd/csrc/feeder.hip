// Host side of the scene feeder (geoformer_amd/feeder.py) in a native thread.
//
// The reference moves a batch with blocking `.cuda()` calls from the driver's thread (test.py:56, train.py:63-75).  The
// feeder of rounds 2-5 staged batch i + 1 into pinned buffers on the CONSUMER's thread at every hand-over: ~0.8 ms of
// memcpy per 150k-point scene during which that thread launches nothing and the device idles (bench.py test_py_shape:
// 5.8 against 5.1 ms per scene; a Python helper thread halved the loop's rate instead -- the interpreter lock).  Here
// the staging memcpy, the asynchronous uploads, the first half of gf_voxelize_idx and the read-back of its two sizes are
// issued by ONE native worker thread per feeder: no interpreter lock is involved (ctypes releases it around the waits
// below), the consumer's thread only builds the job.
//
// Ordering: the worker queues everything of a job on the job's stream in program order and records two events (copies
// done: the pinned buffers may be rewritten; sizes on the host).  The Python side waits for "issued" before it queues
// anything of its own behind the job on that stream (the second half of the voxelisation) -- by then, one hand-over later,
// the worker is long done.
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <string.h>

#include "common.h"

namespace {

constexpr int kSlots = 4;

struct Feeder {
    int device = 0;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<GfFeederJob> queue;
    bool stop = false;
    // per slot: job state (0 free / done, 1 queued or running), status of the last job, its two events
    int state[kSlots] = {0, 0, 0, 0};
    int status[kSlots] = {0, 0, 0, 0};
    char err[kSlots][256];
    hipEvent_t copied[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t head[kSlots] = {nullptr, nullptr, nullptr, nullptr};
};

int run_job(Feeder* f, const GfFeederJob& j) {
    hipStream_t st = (hipStream_t)j.stream;
    for (int i = 0; i < j.n_copies; i++) {
        if (j.bytes[i] == 0) continue;
        memcpy(j.pinned[i], j.src[i], j.bytes[i]);
        GF_TRY(hipMemcpyAsync(j.dev[i], j.pinned[i], j.bytes[i], hipMemcpyHostToDevice, st));
    }
    GF_TRY(hipEventRecord(f->copied[j.slot], st));
    if (j.coords_dev) {
        const int rc = gf_voxelize_idx_count(j.coords_dev, j.N, j.ncol, j.mode, j.scratch, j.input_map, j.head_dev, st);
        if (rc != GF_OK) return rc;
        GF_TRY(hipMemcpyAsync(j.head_host, j.head_dev, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    GF_TRY(hipEventRecord(f->head[j.slot], st));
    return GF_OK;
}

void worker_main(Feeder* f) {
    (void)hipSetDevice(f->device);
    for (;;) {
        GfFeederJob j;
        {
            std::unique_lock<std::mutex> lk(f->mu);
            f->cv.wait(lk, [&] { return f->stop || !f->queue.empty(); });
            if (f->queue.empty()) return;  // (stop, nothing left)
            j = f->queue.front();
            f->queue.pop_front();
        }
        const int rc = run_job(f, j);
        {
            std::lock_guard<std::mutex> lk(f->mu);
            f->status[j.slot] = rc;
            if (rc != GF_OK) {
                strncpy(f->err[j.slot], gf_last_error(), sizeof(f->err[j.slot]) - 1);
                f->err[j.slot][sizeof(f->err[j.slot]) - 1] = 0;
            }
            f->state[j.slot] = 0;
        }
        f->cv.notify_all();
    }
}

}  // namespace

extern "C" void* gf_feeder_create(int device) {
    Feeder* f = new Feeder();
    f->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete f;
        gf_set_error("gf_feeder_create: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    for (int s = 0; s < kSlots; s++) {
        f->err[s][0] = 0;
        if (hipEventCreateWithFlags(&f->copied[s], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f->head[s], hipEventDisableTiming) != hipSuccess) {
            gf_set_error("gf_feeder_create: hipEventCreate failed");
            delete f;
            return nullptr;
        }
    }
    f->worker = std::thread(worker_main, f);
    return f;
}

extern "C" int gf_feeder_submit(void* h, const GfFeederJob* job) {
    Feeder* f = (Feeder*)h;
    GF_CHECK_ARG(f && job, "gf_feeder_submit: null argument");
    GF_CHECK_ARG(job->slot >= 0 && job->slot < kSlots && job->n_copies >= 0 && job->n_copies <= GF_FEEDER_MAX_COPIES,
                 "gf_feeder_submit: slot %d, %d copies", job->slot, job->n_copies);
    {
        std::lock_guard<std::mutex> lk(f->mu);
        GF_CHECK_ARG(f->state[job->slot] == 0, "gf_feeder_submit: slot %d still has a job", job->slot);
        f->state[job->slot] = 1;
        f->status[job->slot] = GF_OK;
        f->queue.push_back(*job);
    }
    f->cv.notify_all();
    return GF_OK;
}

// blocks until the slot's job has been issued to its stream (or failed: its status, message in gf_last_error)
extern "C" int gf_feeder_wait_issued(void* h, int slot) {
    Feeder* f = (Feeder*)h;
    GF_CHECK_ARG(f && slot >= 0 && slot < kSlots, "gf_feeder_wait_issued: bad arguments");
    std::unique_lock<std::mutex> lk(f->mu);
    f->cv.wait(lk, [&] { return f->state[slot] == 0; });
    if (f->status[slot] != GF_OK) gf_set_error("%s", f->err[slot]);
    return f->status[slot];
}

// after gf_feeder_wait_issued: the copies have left the pinned buffers / the voxelisation's sizes are on the host
extern "C" int gf_feeder_wait_copied(void* h, int slot) {
    Feeder* f = (Feeder*)h;
    GF_CHECK_ARG(f && slot >= 0 && slot < kSlots, "gf_feeder_wait_copied: bad arguments");
    GF_TRY(hipEventSynchronize(f->copied[slot]));
    return GF_OK;
}
extern "C" int gf_feeder_wait_head(void* h, int slot) {
    Feeder* f = (Feeder*)h;
    GF_CHECK_ARG(f && slot >= 0 && slot < kSlots, "gf_feeder_wait_head: bad arguments");
    GF_TRY(hipEventSynchronize(f->head[slot]));
    return GF_OK;
}

extern "C" int gf_feeder_destroy(void* h) {
    Feeder* f = (Feeder*)h;
    if (!f) return GF_OK;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->stop = true;
    }
    f->cv.notify_all();
    if (f->worker.joinable()) f->worker.join();
    for (int s = 0; s < kSlots; s++) {
        if (f->copied[s]) (void)hipEventDestroy(f->copied[s]);
        if (f->head[s]) (void)hipEventDestroy(f->head[s]);
    }
    delete f;
    return GF_OK;
}
