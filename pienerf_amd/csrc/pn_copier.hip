// Device-to-host copies of finished frames WITHOUT a HIP stream (gfx950 / ROCm).
//
// The reference copies image, depth and depth_0 to the host every frame (nerf/trainer.py:589-592): 12.8 MB per 800x800 frame, 234 us at the
// PCIe rate (54.7 GB/s measured).  As a hipMemcpyAsync the copy already runs on an SDMA engine (AMD_LOG_LEVEL=4: "HSA Copy copy_engine=0x1";
// the `__amd_rocclr_copyBuffer` kernel that profiles show in its place is what the runtime substitutes while a profiler is attached) — but it
// needs a stream to be ordered on, and this part runs four hardware queues concurrently: on the frame's render stream the copy keeps that
// lane from starting its next frame for 234 us, on a stream of its own it is a fifth busy queue and everything time-slices (measured: 1 155
// steps/s with the copy on the lane, 35 with three lanes + a copy stream, 1 380 without any copy).
//
// Here a host thread takes the place of the stream: it waits for the frame's `render done` event, hands the copy straight to the HSA runtime
// (hsa_amd_memory_async_copy: SDMA, no compute queue involved) and waits for its completion signal.  The render lanes never see the copy.
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "pn_common.h"

struct PnCopyJob {
    void* dst;
    const void* src;
    size_t bytes;
    hipEvent_t after;
    uint64_t ticket;
};

struct pn_copier {
    int device;
    hsa_signal_t signal;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_jobs, cv_done;
    std::deque<PnCopyJob> jobs;
    uint64_t next_ticket, done_ticket;
    int error;  // first failure (hsa_status_t or hipError_t, negated), sticky
    bool stop;
};

static int copy_agents(const void* src_dev, void* dst_host, hsa_agent_t* gpu, hsa_agent_t* cpu) {
    hsa_amd_pointer_info_t info;
    info.size = sizeof(info);
    if (hsa_amd_pointer_info(const_cast<void*>(src_dev), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return -1;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return -2;  // not device memory the HSA runtime knows
    *gpu = info.agentOwner;
    info.size = sizeof(info);
    if (hsa_amd_pointer_info(dst_host, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return -3;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA && info.type != HSA_EXT_POINTER_TYPE_LOCKED) return -4;  // not pinned host memory
    *cpu = info.agentOwner;
    return 0;
}

static void copier_loop(pn_copier* c) {
    (void)hipSetDevice(c->device);
    for (;;) {
        PnCopyJob j;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv_jobs.wait(lk, [&] { return c->stop || !c->jobs.empty(); });
            if (c->jobs.empty()) return;  // stop requested and nothing left
            j = c->jobs.front();
            c->jobs.pop_front();
        }
        int err = 0;
        if (j.after) {
            const hipError_t e = hipEventSynchronize(j.after);
            if (e != hipSuccess) err = -(int)e;
        }
        hsa_agent_t gpu, cpu;
        if (!err) {
            const int a = copy_agents(j.src, j.dst, &gpu, &cpu);
            if (a) err = -1000 + a;
        }
        if (!err) {
            hsa_signal_store_relaxed(c->signal, 1);
            const hsa_status_t s = hsa_amd_memory_async_copy(j.dst, cpu, j.src, gpu, j.bytes, 0, nullptr, c->signal);
            if (s != HSA_STATUS_SUCCESS) err = -2000 - (int)s;
            else if (hsa_signal_wait_scacquire(c->signal, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) < 0) err = -3000;
        }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (err && !c->error) c->error = err;
            c->done_ticket = j.ticket;
        }
        c->cv_done.notify_all();
    }
}

extern "C" int pn_copier_create(pn_copier** out) {
    PN_REQUIRE(out);
    pn_copier* c = new pn_copier();
    c->next_ticket = 1; c->done_ticket = 0; c->error = 0; c->stop = false;
    PN_HIP_CHECK(hipGetDevice(&c->device));
    if (hsa_init() != HSA_STATUS_SUCCESS) { delete c; snprintf(pn_err_buf, sizeof(pn_err_buf), "pn_copier_create: hsa_init failed"); return PN_ERR_HIP; }  // reference-counted: HIP holds the runtime already
    if (hsa_signal_create(1, 0, nullptr, &c->signal) != HSA_STATUS_SUCCESS) {
        (void)hsa_shut_down(); delete c; snprintf(pn_err_buf, sizeof(pn_err_buf), "pn_copier_create: hsa_signal_create failed"); return PN_ERR_HIP;
    }
    c->worker = std::thread(copier_loop, c);
    *out = c;
    return PN_OK;
}

extern "C" void pn_copier_destroy(pn_copier* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv_jobs.notify_all();
    if (c->worker.joinable()) c->worker.join();
    (void)hsa_signal_destroy(c->signal);
    (void)hsa_shut_down();
    delete c;
}

extern "C" int pn_copier_submit(pn_copier* c, void* dst_host, const void* src_dev, uint64_t bytes, void* after_event, uint64_t* ticket) {
    PN_REQUIRE(c && dst_host && src_dev && bytes > 0 && ticket);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        *ticket = c->next_ticket++;
        c->jobs.push_back(PnCopyJob{dst_host, src_dev, (size_t)bytes, (hipEvent_t)after_event, *ticket});
    }
    c->cv_jobs.notify_one();
    return PN_OK;
}

extern "C" int pn_copier_wait(pn_copier* c, uint64_t ticket) {
    PN_REQUIRE(c && ticket > 0);
    std::unique_lock<std::mutex> lk(c->mu);
    PN_REQUIRE(ticket < c->next_ticket);
    c->cv_done.wait(lk, [&] { return c->done_ticket >= ticket; });
    if (c->error) {
        snprintf(pn_err_buf, sizeof(pn_err_buf), "pn_copier: a copy failed with code %d (-1..-999: HIP event, -10xx: pointer lookup, -2xxx: hsa_amd_memory_async_copy, -3000: signal)", c->error);
        return PN_ERR_HIP;
    }
    return PN_OK;
}
