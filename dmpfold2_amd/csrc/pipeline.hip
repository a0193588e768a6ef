// Throughput mode behind the C ABI (round 6; include/dmpfold_hip.h, dmp_pipeline_*): S contexts on S HIP streams share a
// lane, and ONE host thread inside the library schedules all of them unit by unit - the scheduler that lived in Python
// (dmpfold2_amd.predict.Pipeline._pump, rounds 1-5), statement for statement: light units are enqueued at once, a residual
// block - whose convolution takes the lane, in issue order - only when everything its engine was given has completed, so
// that the lane is always handed to a convolution that can start immediately; targets that begin together run their
// vertical GRUs as one chain, which also serves the next targets of the queue as riders.  The caller submits device
// buffers and polls / waits for tickets; nothing of the issue loop runs in the caller's language.
//
// Replaces, in the reference, nothing: predict.py:74-158 predicts one alignment per call.  It is the batch form of that
// call - N independent alignments through one GPU - which a maintainer of the reference would bind for throughput
// (INTEGRATION.md section 3).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <pthread.h>
#include <time.h>

using namespace dmp;

namespace {

struct Job {
  int64_t ticket;
  const uint8_t* msa;
  int N, L;
  const float* tpl;
  int nloops, refine;
  float* coords;
  float* conf;
  void* ready;             // hipEvent_t recorded by the caller behind the producer of msa / tpl (or null)
};

enum TicketState { T_QUEUED = 0, T_RUNNING = 1, T_ISSUED = 2, T_DONE = 3, T_FAILED = 4 };

struct Ticket {
  int state = T_QUEUED;
  int rc = 0;              // DMP_* of a failed begin / issue
  int slot = -1;           // fault slot (pinned host word the latch kernel writes)
  hipEvent_t done = nullptr;
  bool reported = false;   // handed out by dmp_pipeline_poll
};

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Engine streams come from a process-wide pool per device that is made 32 streams at a time - the way PyTorch makes its
// stream pool, which the Python scheduler of rounds 1-5 drew its engine streams from.  It matters: with only the four
// engine streams in existence the runtime (ROCm 7.2, GPU_MAX_HW_QUEUES=8) lets some of them share a hardware queue, the
// lane's two convolutions then run one behind the other more often (launches in flight 1.65 instead of 1.90) and the
// scheduler loses 6 % (7.05 against 7.52 structures/s in the fast mode, 4.00 against 4.11 in precision 2; alternating
// runs on one box, profiles/r06_scheduler_c_abi_ab.txt); whether the streams are made with hipStreamCreateWithFlags or
// ...WithPriority makes no difference.  Streams are reused from one pipeline of a process to the next (rounds 3-5: a
// pipeline on the pool's streams 5-8 was 15 % slower than one on 1-4, tools/pipeline_order.py).
std::mutex g_pool_mu;
std::map<int, std::vector<std::pair<hipStream_t, bool>>> g_stream_pool;

int take_stream(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  auto& pool = g_stream_pool[device];
  for (auto& it : pool)
    if (!it.second) { it.second = true; *out = it.first; return DMP_OK; }
  const size_t first = pool.size();
  for (int i = 0; i < 32; ++i) {
    hipStream_t x;
    DMP_HIP(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    pool.push_back({x, false});
  }
  pool[first].second = true;
  *out = pool[first].first;
  return DMP_OK;
}
void release_stream(int device, hipStream_t st) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  for (auto& it : g_stream_pool[device])
    if (it.first == st) it.second = false;
}

}  // namespace

struct dmp_pipeline {
  int device = 0, max_L = 0, max_N = 0;
  bool own_streams = true;                 // false: the caller's streams (dmp_pipeline_create_on)
  std::vector<dmp_ctx*> ctx;
  std::vector<hipStream_t> stream;
  // scheduler knobs (the environment variables of the Python scheduler, read once)
  int tail_stagger = 8, group_max = 4, group_patience = 40, riders_max = 4;
  // ---- shared with the callers (mu) ----
  std::mutex mu;
  std::condition_variable cv_work, cv_state;
  std::deque<Job> pending;
  std::map<int64_t, Ticket> tickets;
  int64_t next_ticket = 0;
  bool stop = false;
  bool paused = false;                     // dmp_pipeline_pause: no NEW target is started (the running ones go on)
  int running = 0;                         // engines with a prediction in flight (issue not finished)
  char last_error[512] = "";
  // ---- scheduler thread only ----
  std::thread thread;
  struct Slot { bool busy = false; Job job; int done = 0, total = 0; int ahead_buf = -1; hipEvent_t ahead_ev = nullptr; };
  std::vector<Slot> slot;
  struct RiderWait { std::vector<Job> jobs; std::vector<int> bufs; };
  std::vector<RiderWait> rider_wait;       // per leading engine
  std::set<int64_t> riding;                // tickets whose chain has not been issued to its end yet
  struct Ahead { int buf; hipEvent_t ev; };
  std::map<int64_t, Ahead> ahead;          // ticket -> vertical-GRU result that rode in an earlier chain
  std::vector<float*> rider_buf;           // pool of [max_L][512] device buffers
  std::vector<bool> rider_buf_used;
  std::mutex ev_mu;                        // the event pool is fed by dmp_pipeline_release (callers) too
  std::vector<hipEvent_t> ev_pool;
  int* fault_words = nullptr;              // pinned host memory, one word per fault slot
  std::vector<bool> fault_slot_used;
  std::atomic<long long> stat_rider_chains{0}, stat_max_riders{0}, stat_max_group{0}, stat_groups{0}, stat_idle_rounds{0};
  std::atomic<long long> stat_thread_cpu_us{0}, stat_rounds{0};
};

namespace {

hipEvent_t take_event(dmp_pipeline* p) {
  {
    std::lock_guard<std::mutex> g(p->ev_mu);
    if (!p->ev_pool.empty()) { hipEvent_t e = p->ev_pool.back(); p->ev_pool.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

int take_rider_buf(dmp_pipeline* p) {
  for (size_t i = 0; i < p->rider_buf.size(); ++i)
    if (!p->rider_buf_used[i]) { p->rider_buf_used[i] = true; return (int)i; }
  float* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(float) * (size_t)p->max_L * WIDTH) != hipSuccess) return -1;
  p->rider_buf.push_back(d);
  p->rider_buf_used.push_back(true);
  return (int)p->rider_buf.size() - 1;
}

void give_event(dmp_pipeline* p, hipEvent_t e) {
  if (!e) return;
  std::lock_guard<std::mutex> g(p->ev_mu);
  p->ev_pool.push_back(e);
}

void fail_ticket(dmp_pipeline* p, int64_t t, int rc) {
  std::lock_guard<std::mutex> g(p->mu);
  Ticket& tk = p->tickets[t];
  tk.state = T_FAILED;
  tk.rc = rc;
  snprintf(p->last_error, sizeof(p->last_error), "ticket %lld: %s", (long long)t, dmp_last_error());
  p->cv_state.notify_all();
}

// start job on engine s (Pipeline._begin)
int begin(dmp_pipeline* p, int s, const Job& job) {
  hipStream_t st = p->stream[s];
  if (job.ready) DMP_HIP(hipStreamWaitEvent(st, (hipEvent_t)job.ready, 0));
  int rc = dmp_predict_begin_units(p->ctx[s], job.msa, job.N, job.L, job.tpl, job.tpl ? job.L : 0, job.nloops, job.refine);
  if (rc) return rc;
  dmp_pipeline::Slot& sl = p->slot[s];
  sl.busy = true;
  sl.job = job;
  sl.done = 0;
  sl.total = (job.nloops + 1) * NBLOCK;
  sl.ahead_buf = -1;
  sl.ahead_ev = nullptr;
  return DMP_OK;
}

// Start the next slots.size() queued targets on these free engines; those whose vertical GRU did not ride in an earlier
// chain form one vertical-GRU group (the first of them leads), and the chain takes the next queued targets along as riders
// (Pipeline._begin_group).  Called with p->mu NOT held; takes it for the queue.
void begin_group(dmp_pipeline* p, const std::vector<int>& slots) {
  std::vector<int> grouped;
  for (int s : slots) {
    Job job;
    {
      std::lock_guard<std::mutex> g(p->mu);
      if (p->pending.empty()) break;
      job = p->pending.front();
      p->pending.pop_front();
      p->tickets[job.ticket].state = T_RUNNING;
      p->running++;
      p->cv_state.notify_all();
    }
    int rc = begin(p, s, job);
    if (rc) {
      { std::lock_guard<std::mutex> g(p->mu); p->running--; }
      fail_ticket(p, job.ticket, rc);
      continue;
    }
    auto it = p->ahead.find(job.ticket);
    if (it != p->ahead.end()) {
      rc = dmp_predict_set_vgru_result(p->ctx[s], p->rider_buf[it->second.buf], (void*)it->second.ev);
      // buffer and event go back to their pools when this prediction has been issued to its end: the wait for the event is
      // enqueued in this engine's last front-end unit, and an event must not be re-recorded before every wait for it has
      // been enqueued (a later record that completes earlier would release the waiter too soon)
      p->slot[s].ahead_buf = it->second.buf;
      p->slot[s].ahead_ev = it->second.ev;
      p->ahead.erase(it);
      if (rc) {
        p->rider_buf_used[p->slot[s].ahead_buf] = false;
        give_event(p, p->slot[s].ahead_ev);
        p->slot[s].ahead_buf = -1; p->slot[s].ahead_ev = nullptr;
        p->slot[s].busy = false;
        { std::lock_guard<std::mutex> g(p->mu); p->running--; }
        fail_ticket(p, job.ticket, rc);
      }
    } else {
      grouped.push_back(s);
    }
  }
  if (grouped.size() > 1) {
    dmp_ctx* ctxs[8];
    for (size_t i = 0; i < grouped.size(); ++i) ctxs[i] = p->ctx[grouped[i]];
    if (dmp_predict_group_vgru(ctxs, (int)grouped.size()) != DMP_OK) return;     // (members then run their own chains)
    p->stat_groups++;
    if ((long long)grouped.size() > p->stat_max_group) p->stat_max_group = (long long)grouped.size();
    if (p->riders_max > 0) {
      const int lead = grouped[0];
      const int room = std::min(p->riders_max, 8 - (int)grouped.size());
      std::vector<Job> jobs;
      {
        std::lock_guard<std::mutex> g(p->mu);
        for (int i = 0; i < room && i < (int)p->pending.size(); ++i) {
          const Job& j = p->pending[i];
          if (!p->ahead.count(j.ticket) && !p->riding.count(j.ticket)) jobs.push_back(j);
        }
      }
      if (!jobs.empty()) {
        const int k = (int)jobs.size();
        const uint8_t* mp[8];
        float* op[8];
        int Ns[8], Ls[8];
        std::vector<int> bufs;
        bool ok = true;
        for (int i = 0; i < k && ok; ++i) {
          const int b = take_rider_buf(p);
          if (b < 0) { ok = false; break; }
          bufs.push_back(b);
          if (jobs[i].ready && hipStreamWaitEvent(p->stream[lead], (hipEvent_t)jobs[i].ready, 0) != hipSuccess) ok = false;
          mp[i] = jobs[i].msa; op[i] = p->rider_buf[b]; Ns[i] = jobs[i].N; Ls[i] = jobs[i].L;
        }
        if (ok && dmp_predict_group_riders(p->ctx[lead], k, mp, Ns, Ls, op) == DMP_OK) {
          p->rider_wait[lead].jobs = jobs;
          p->rider_wait[lead].bufs = bufs;
          p->stat_rider_chains++;
          if (k > p->stat_max_riders) p->stat_max_riders = k;
          for (const Job& j : jobs) p->riding.insert(j.ticket);
        } else {
          for (int b : bufs) p->rider_buf_used[b] = false;
        }
      }
    }
  }
}

// One scheduling round over the engines; true if anything was enqueued (Pipeline._pump).
bool pump(dmp_pipeline* p) {
  const int S = (int)p->ctx.size();
  const bool gated = S > 1;
  bool progressed = false;
  int n_conv = 0;
  if (gated)
    for (int s = 0; s < S; ++s)
      if (p->slot[s].busy && dmp_predict_next_unit(p->ctx[s]) == 2) ++n_conv;
  std::vector<int> free_slots;
  for (int s = 0; s < S; ++s)
    if (!p->slot[s].busy) free_slots.push_back(s);
  size_t n_pending;
  bool rider_first = false;
  {
    std::lock_guard<std::mutex> g(p->mu);
    n_pending = p->paused ? 0 : p->pending.size();
    const size_t k = std::min(free_slots.size(), n_pending);
    for (size_t i = 0; i < k; ++i)
      if (p->riding.count(p->pending[i].ticket)) rider_first = true;     // its chain has not been issued to its end yet
  }
  std::vector<int> startable(free_slots.begin(), free_slots.begin() + std::min(free_slots.size(), n_pending));
  if (rider_first) startable.clear();
  if (!startable.empty()) {
    // engines about to finish: wait for them and start together (one vertical-GRU chain for the group)
    int soon = 0;
    for (int r = 0; r < S; ++r)
      if (p->slot[r].busy && p->slot[r].total - p->slot[r].done <= p->group_patience) ++soon;
    const int soon_with_work = std::min((int)n_pending - (int)startable.size(), soon);
    const int want = std::min(p->group_max, (int)startable.size() + soon_with_work);
    if ((int)startable.size() >= want || !soon_with_work || p->group_max == 1) {
      if (p->group_max == 1) {
        for (int s : startable) begin_group(p, {s});
      } else {
        startable.resize(std::max(want, 1));
        begin_group(p, startable);
      }
      progressed = true;
    }
  }
  for (int s = 0; s < S; ++s) {
    dmp_pipeline::Slot& sl = p->slot[s];
    if (!sl.busy) continue;
    dmp_ctx* c = p->ctx[s];
    hipStream_t st = p->stream[s];
    while (true) {
      const int kind = dmp_predict_next_unit(c);
      if (kind == 3) break;                                 // waits for its group leader's vertical-GRU chain
      int rc = DMP_OK;
      if (kind == 0) {
        // final refinement + backbone; neither needs the lane
        int fslot = -1;
        hipEvent_t ev;
        {
          std::lock_guard<std::mutex> g(p->mu);
          for (size_t i = 0; i < p->fault_slot_used.size(); ++i)
            if (!p->fault_slot_used[i]) { fslot = (int)i; p->fault_slot_used[i] = true; break; }
        }
        ev = take_event(p);
        if (fslot >= 0) p->fault_words[fslot] = 0;
        c->end_fault_out = fslot >= 0 ? &p->fault_words[fslot] : nullptr;
        rc = dmp_predict_end(c, sl.job.coords, sl.job.conf, (void*)st);
        c->end_fault_out = nullptr;
        if (!rc && ev && hipEventRecord(ev, st) != hipSuccess) rc = DMP_ERR_HIP;
        if (sl.ahead_buf >= 0) { p->rider_buf_used[sl.ahead_buf] = false; sl.ahead_buf = -1; }
        give_event(p, sl.ahead_ev);
        sl.ahead_ev = nullptr;
        sl.busy = false;
        {
          std::lock_guard<std::mutex> g(p->mu);
          Ticket& tk = p->tickets[sl.job.ticket];
          tk.slot = fslot;
          tk.done = ev;
          tk.state = rc ? T_FAILED : T_ISSUED;
          tk.rc = rc;
          p->running--;
          p->cv_state.notify_all();
        }
        progressed = true;
        break;
      }
      if (p->tail_stagger && kind == 2 && sl.done == 0) {
        // An engine issues its first residual block only when the engine that started before it is half a pass into its
        // own: engines that leave their front ends together would otherwise reach the end of every pass together and
        // sit in their pass tails (eigensolver, coordinate GRU: no convolution) at the same time.
        int prev = -1;
        for (int r = 0; r < S; ++r)
          if (r != s && p->slot[r].busy && p->slot[r].job.ticket < sl.job.ticket &&
              (prev < 0 || p->slot[r].job.ticket > p->slot[prev].job.ticket)) prev = r;
        if (prev >= 0) {
          const bool in_trunk = p->slot[prev].done > 0 || dmp_predict_next_unit(p->ctx[prev]) == 2;
          if (in_trunk && p->slot[prev].done < std::min(p->tail_stagger, p->slot[prev].total)) break;
        }
      }
      if (gated) {
        // a convolution is handed the lane only when it can start at once; light units are kept one deep so that this
        // loop returns to the other engines quickly
        const int busy = dmp_ctx_pending(c);
        if (busy < 0) rc = busy;
        else if (busy > ((kind == 2 && n_conv > 1) ? 0 : 1)) break;
      }
      if (!rc) rc = dmp_predict_issue_unit(c, (void*)st);
      if (rc) {
        // this prediction cannot go on: its ticket fails, the engine is free again (its group's members, if it led a
        // chain, would wait for ever: they fail with it)
        sl.busy = false;
        { std::lock_guard<std::mutex> g(p->mu); p->running--; }
        fail_ticket(p, sl.job.ticket, rc);
        for (int r = 0; r < S; ++r)
          if (r != s && p->slot[r].busy && p->ctx[r]->vg_leader == c) {
            p->ctx[r]->vg_leader = nullptr;
            p->slot[r].busy = false;
            { std::lock_guard<std::mutex> g(p->mu); p->running--; }
            fail_ticket(p, p->slot[r].job.ticket, rc);
          }
        c->vg_waiters = 0;
        progressed = true;
        break;
      }
      progressed = true;
      if (!p->rider_wait[s].jobs.empty()) {
        int issued = 0;
        dmp_ctx_get_option(c, "chain_issued", &issued);
        if (issued) {
          // the riders' results are behind this point of the leader's stream
          hipEvent_t ev = take_event(p);
          if (ev) (void)hipEventRecord(ev, st);
          for (size_t i = 0; i < p->rider_wait[s].jobs.size(); ++i) {
            // one event per rider (each is returned to the pool by the engine that consumes that rider's result)
            hipEvent_t e = i == 0 ? ev : take_event(p);
            if (i > 0 && e) (void)hipEventRecord(e, st);
            p->ahead[p->rider_wait[s].jobs[i].ticket] = {p->rider_wait[s].bufs[i], e};
          }
          {
            std::lock_guard<std::mutex> g(p->mu);
            for (const Job& j : p->rider_wait[s].jobs) p->riding.erase(j.ticket);
          }
          p->rider_wait[s].jobs.clear();
          p->rider_wait[s].bufs.clear();
        }
      }
      if (kind == 2) sl.done++;
      if (gated) break;
    }
  }
  return progressed;
}

// completion sweep: tickets whose done event has fired become T_DONE (scheduler thread and waiters)
void sweep_done(dmp_pipeline* p) {
  std::lock_guard<std::mutex> g(p->mu);
  bool any = false;
  for (auto& kv : p->tickets) {
    Ticket& tk = kv.second;
    if (tk.state == T_ISSUED && tk.done && hipEventQuery(tk.done) == hipSuccess) { tk.state = T_DONE; any = true; }
  }
  if (any) p->cv_state.notify_all();
}

void scheduler_main(dmp_pipeline* p) {
  (void)pthread_setname_np(pthread_self(), "dmp-scheduler");
  (void)hipSetDevice(p->device);
  while (true) {
    {
      std::unique_lock<std::mutex> lk(p->mu);
      // nothing queued, nothing in flight: sleep until a submit (or the end)
      p->cv_work.wait(lk, [&] {
        if (p->stop || (!p->pending.empty() && !p->paused) || p->running > 0) return true;
        for (auto& kv : p->tickets)
          if (kv.second.state == T_ISSUED) return true;
        return false;
      });
      if (p->stop) return;
    }
    const bool progressed = pump(p);
    sweep_done(p);
    if ((++p->stat_rounds & 255) == 0) {
      timespec ts;
      if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0) p->stat_thread_cpu_us = (long long)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
    }
    if (!progressed) {
      // Nothing could be issued: every engine waits for the GPU (or for an engine that does).  A bounded 20 us sleep per
      // idle round (round 3 measured: costs the throughput nothing; 100 us: 1.3 %; blocking on the oldest outstanding
      // unit: a fifth of the throughput - units complete out of order across the engines).  With nothing left to ISSUE
      // (only completions outstanding) the thread waits longer: nobody needs it until the next submit.
      bool only_completions;
      {
        std::lock_guard<std::mutex> g(p->mu);
        only_completions = p->pending.empty() && p->running == 0;
      }
      p->stat_idle_rounds++;
      std::this_thread::sleep_for(std::chrono::microseconds(only_completions ? 200 : 20));
    }
  }
}

}  // namespace

extern "C" {

int dmp_pipeline_create_on(int device, int max_L, int max_N, int engines, void* const* streams, dmp_pipeline** out);

int dmp_pipeline_create(int device, int max_L, int max_N, int engines, dmp_pipeline** out) {
  return dmp_pipeline_create_on(device, max_L, max_N, engines, nullptr, out);
}

int dmp_pipeline_create_on(int device, int max_L, int max_N, int engines, void* const* streams, dmp_pipeline** out) {
  DMP_ARG(out != nullptr, "out is NULL");
  DMP_ARG(engines >= 1 && engines <= 8, "a pipeline has 1..8 engines, got %d", engines);
  DMP_HIP(hipSetDevice(device));
  dmp_pipeline* p = new dmp_pipeline();
  p->device = device;
  p->max_L = max_L;
  p->max_N = max_N;
  p->own_streams = streams == nullptr;
  p->tail_stagger = env_int("DMP_TAIL_STAGGER", 8);
  p->group_max = std::max(1, std::min(4, env_int("DMP_VGRU_GROUP", 4)));
  p->group_patience = env_int("DMP_GROUP_PATIENCE", 40);
  p->riders_max = engines > 1 ? std::max(0, std::min(7, env_int("DMP_VGRU_RIDERS", 4))) : 0;
  int rc = DMP_OK;
  for (int i = 0; i < engines && !rc; ++i) {
    hipStream_t st;
    if (streams) st = (hipStream_t)streams[i];
    else if ((rc = take_stream(device, &st))) break;
    p->stream.push_back(st);
    dmp_ctx* c = nullptr;
    if ((rc = dmp_ctx_create(device, max_L, max_N, &c))) break;
    p->ctx.push_back(c);
    if (engines > 1) {
      if (i > 0 && (rc = dmp_ctx_share_lane(c, p->ctx[0]))) break;       // one lane for all of them
      // Several engines: the eigensolver's Householder steps as one launch each, not as the cluster kernel (same bits on
      // full-rank matrices).  The cluster is the faster form for ONE prediction, but its 32 resident workgroups poll
      // beside the other engines' convolutions for 0.9 ms per pass: 7.39 / 7.32 structures/s with the launches against
      // 7.29 / 7.30 (round 3).
      dmp_ctx_set_option(c, "tridiag_cluster", 0);
    }
  }
  if (!rc && hipHostMalloc((void**)&p->fault_words, sizeof(int) * 256, hipHostMallocMapped) != hipSuccess) rc = DMP_ERR_HIP;
  if (rc) { dmp_pipeline_destroy(p); return rc; }
  p->fault_slot_used.assign(256, false);
  p->slot.resize(engines);
  p->rider_wait.resize(engines);
  p->thread = std::thread(scheduler_main, p);
  *out = p;
  return DMP_OK;
}

void dmp_pipeline_destroy(dmp_pipeline* p) {
  if (!p) return;
  if (p->thread.joinable()) {
    { std::lock_guard<std::mutex> g(p->mu); p->stop = true; }
    p->cv_work.notify_all();
    p->thread.join();
  }
  (void)hipSetDevice(p->device);
  for (hipStream_t st : p->stream) (void)hipStreamSynchronize(st);
  for (dmp_ctx* c : p->ctx) dmp_ctx_destroy(c);
  if (p->own_streams)
    for (hipStream_t st : p->stream) release_stream(p->device, st);
  for (auto& kv : p->tickets)
    if (kv.second.done) (void)hipEventDestroy(kv.second.done);
  for (auto& kv : p->ahead)
    if (kv.second.ev) (void)hipEventDestroy(kv.second.ev);
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  for (float* d : p->rider_buf) (void)hipFree(d);
  if (p->fault_words) (void)hipHostFree(p->fault_words);
  delete p;
}

int dmp_pipeline_engines(const dmp_pipeline* p) { return p ? (int)p->ctx.size() : 0; }

dmp_ctx* dmp_pipeline_ctx(dmp_pipeline* p, int i) {
  return (p && i >= 0 && i < (int)p->ctx.size()) ? p->ctx[i] : nullptr;
}

void* dmp_pipeline_stream(dmp_pipeline* p, int i) {
  return (p && i >= 0 && i < (int)p->stream.size()) ? (void*)p->stream[i] : nullptr;
}

int dmp_pipeline_weights_ready(dmp_pipeline* p) {
  DMP_ARG(p != nullptr, "null pipeline");
  for (size_t i = 1; i < p->ctx.size(); ++i) {
    int rc = dmp_weights_share(p->ctx[i], p->ctx[0]);           // packed once per pipeline, not once per engine
    if (rc) return rc;
  }
  return DMP_OK;
}

int dmp_pipeline_set_option(dmp_pipeline* p, const char* name, int value) {
  DMP_ARG(p && name, "null argument");
  {
    std::lock_guard<std::mutex> g(p->mu);
    DMP_ARG(p->pending.empty() && p->running == 0, "options are set on an idle pipeline (a group's chain runs in its leader's "
            "arithmetic and serves every member)");
  }
  for (dmp_ctx* c : p->ctx) {
    int rc = dmp_ctx_set_option(c, name, value);
    if (rc) return rc;
  }
  return DMP_OK;
}

int64_t dmp_pipeline_submit(dmp_pipeline* p, const uint8_t* d_msa, int N, int L, const float* d_template_ca, int nloops,
                            int refine_steps, float* d_coords, float* d_conf, void* ready_event) {
  DMP_ARG(p && d_msa && d_coords && d_conf, "null argument");
  DMP_ARG(N >= 1 && L >= 8, "need N >= 1 and L >= 8 (got N=%d L=%d)", N, L);
  if (L > p->max_L || N > p->max_N) {
    set_error("alignment %d x %d exceeds the pipeline capacity %d x %d", N, L, p->max_N, p->max_L);
    return DMP_ERR_CAPACITY;
  }
  DMP_ARG(p->ctx[0]->W.ready, "weights not finalized (dmp_weights_set / dmp_weights_finalize on dmp_pipeline_ctx(p, 0), then "
          "dmp_pipeline_weights_ready)");
  int64_t t;
  {
    std::lock_guard<std::mutex> g(p->mu);
    t = p->next_ticket++;
    p->tickets[t] = Ticket();
    p->pending.push_back({t, d_msa, N, L, d_template_ca, nloops < 0 ? 0 : nloops, refine_steps < 0 ? 0 : refine_steps, d_coords,
                          d_conf, ready_event});
  }
  p->cv_work.notify_all();
  return t;
}

int dmp_pipeline_wait(dmp_pipeline* p, int what) {
  DMP_ARG(p != nullptr && what >= 0 && what <= 2, "what: 0 = started, 1 = issued, 2 = completed");
  std::unique_lock<std::mutex> lk(p->mu);
  DMP_ARG(!p->paused || p->pending.empty(), "the pipeline is paused with targets queued (dmp_pipeline_pause(p, 0) first)");
  if (what == 0) {
    p->cv_state.wait(lk, [&] { return p->pending.empty(); });
    return DMP_OK;
  }
  p->cv_state.wait(lk, [&] { return p->pending.empty() && p->running == 0; });
  if (what == 1) return DMP_OK;
  // completed: synchronise with the done events of everything issued (outside the lock: the scheduler keeps running)
  std::vector<hipEvent_t> evs;
  for (auto& kv : p->tickets)
    if (kv.second.state == T_ISSUED && kv.second.done) evs.push_back(kv.second.done);
  lk.unlock();
  for (hipEvent_t e : evs) DMP_HIP(hipEventSynchronize(e));
  sweep_done(p);
  return DMP_OK;
}

int dmp_pipeline_poll(dmp_pipeline* p, int64_t* h_tickets, int capacity, int* h_n) {
  DMP_ARG(p && h_tickets && h_n && capacity >= 0, "bad argument");
  sweep_done(p);
  std::lock_guard<std::mutex> g(p->mu);
  int n = 0;
  for (auto& kv : p->tickets) {
    Ticket& tk = kv.second;
    if (n >= capacity) break;
    if ((tk.state == T_DONE || tk.state == T_FAILED) && !tk.reported) { tk.reported = true; h_tickets[n++] = kv.first; }
  }
  *h_n = n;
  return DMP_OK;
}

int dmp_pipeline_status(dmp_pipeline* p, int64_t ticket, int* h_state, int* h_fault_bits) {
  DMP_ARG(p && h_state, "null argument");
  sweep_done(p);
  std::lock_guard<std::mutex> g(p->mu);
  auto it = p->tickets.find(ticket);
  DMP_ARG(it != p->tickets.end(), "unknown ticket %lld", (long long)ticket);
  *h_state = it->second.state;
  if (h_fault_bits) *h_fault_bits = (it->second.state == T_DONE && it->second.slot >= 0) ? p->fault_words[it->second.slot] : 0;
  if (it->second.state == T_FAILED) {
    set_error("%s", p->last_error);
    return it->second.rc ? it->second.rc : DMP_ERR_FAULT;
  }
  return DMP_OK;
}

int dmp_pipeline_release(dmp_pipeline* p, int64_t ticket) {
  DMP_ARG(p != nullptr, "null pipeline");
  std::lock_guard<std::mutex> g(p->mu);
  auto it = p->tickets.find(ticket);
  DMP_ARG(it != p->tickets.end(), "unknown ticket %lld", (long long)ticket);
  DMP_ARG(it->second.state == T_DONE || it->second.state == T_FAILED, "ticket %lld is still in flight", (long long)ticket);
  if (it->second.slot >= 0) p->fault_slot_used[it->second.slot] = false;
  give_event(p, it->second.done);
  p->tickets.erase(it);
  return DMP_OK;
}

int dmp_pipeline_pause(dmp_pipeline* p, int on) {
  DMP_ARG(p != nullptr, "null pipeline");
  { std::lock_guard<std::mutex> g(p->mu); p->paused = on != 0; }
  p->cv_work.notify_all();
  return DMP_OK;
}

int dmp_pipeline_stats(dmp_pipeline* p, long long* h_stats, int capacity) {
  DMP_ARG(p && h_stats && capacity >= 0, "bad argument");
  long long ahead_left;
  {
    // (the scheduler's own containers: read while it is idle - after dmp_pipeline_wait(p, 1) - for an exact answer)
    std::lock_guard<std::mutex> g(p->mu);
    ahead_left = (long long)p->ahead.size() + (long long)p->riding.size();
  }
  const long long v[8] = {p->stat_groups, p->stat_max_group, p->stat_rider_chains, p->stat_max_riders, ahead_left, p->stat_idle_rounds,
                          p->stat_rounds, p->stat_thread_cpu_us};
  for (int i = 0; i < capacity && i < 8; ++i) h_stats[i] = v[i];
  return DMP_OK;
}

int dmp_pipeline_backlog(dmp_pipeline* p, int* h_queued, int* h_running) {
  DMP_ARG(p && h_queued && h_running, "null argument");
  std::lock_guard<std::mutex> g(p->mu);
  *h_queued = (int)p->pending.size();
  *h_running = p->running;
  return DMP_OK;
}

}  // extern "C"
