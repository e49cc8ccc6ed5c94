// comm.h — inter-rank transport of the decomposed engine (see comm.cpp)
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "device.h"

namespace lmp_le {

struct Msg { void *dev; size_t bytes; int peer; };

struct Comm {
  enum Backend { NONE, RCCL, SHM, LOCAL } backend = NONE;
  // set by abort(): the communicator is gone (ncclCommAbort) or poisoned (test transports: the peers were told to stop
  // waiting); from then on every collective of this handle throws `communicator aborted ...` instead of quietly doing
  // nothing or a one-rank version of itself, and Engine::run refuses to start.  world stays what it was.
  bool aborted = false;
  void require_alive() const;
  int rank = 0, world = 1;
  // RCCL calls are issued on the engine's stream (one total order per rank); the opt-in halo/compute overlap
  // (LAMMPS_LE_OVERLAP) additionally issues the per-step halo on comm_stream, ordered against the main stream by events
  hipStream_t main_stream = nullptr;
  // A rank that stops on an error (its own LammpsError) leaves its peers inside a collective that can never complete.
  // Every host-side wait of a decomposed RCCL run therefore goes through wait_stream(): it polls the stream and the
  // communicator's asynchronous error state and, after timeout_s (LAMMPS_LE_COMM_TIMEOUT, default 120 s - what the test
  // transports use), aborts the communicator and throws, so that every rank ends with an error instead of hanging.
  double timeout_s = 120.0;
  void wait_stream(hipStream_t st);
  // called by a rank that leaves a run with an error: ncclCommAbort for RCCL; the test transports raise a flag their
  // peers' waits look at (hub flag / marker file), so that those end at once with the same message instead of timing out
  void abort();
  void init(const std::string &backend_name, int rank, int world, const void *unique_id, const std::string &session);
  void finalize();
  int nranks();      // size of the communicator as the transport itself reports it (ncclCommCount for RCCL)
  // device buffers
  void exchange(hipStream_t st, const std::vector<Msg> &sends, const std::vector<Msg> &recvs);
  void allreduce_int_max(hipStream_t st, int *dev, int n);
  // element-wise sum of n 32-bit words; used for bit masks whose bits have exactly one owner each (sum = OR, no carries)
  void allreduce_u32_sum(hipStream_t st, unsigned *dev, size_t n);
  // bytes this rank contributed to device collectives since init (what a firing of the LE fixes moves: lammps_le_stat)
  double bytes_allgather = 0.0, bytes_allreduce = 0.0;
  void allgather(hipStream_t st, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
  // host buffers (rebuild-time counts, thermo sums)
  void exchange_host(const std::vector<Msg> &sends, const std::vector<Msg> &recvs);
  void allgather_host(const void *send, void *recv, size_t bytes_per_rank);
  double allreduce_host_sum(double v);
  void allreduce_host_sum(double *v, int n);
  long allreduce_host_max(long v);
  void barrier();

 private:
  void *rccl_comm = nullptr;             // ncclComm_t of THIS engine instance
  std::shared_ptr<struct LocalHub> hub;   // backend "local": ranks are engine instances (threads) of one process
  std::string shm_dir;
  std::vector<long> shm_sent, shm_rcvd;
  void shm_send(int dst, const void *buf, size_t bytes);
  void shm_recv(int src, void *buf, size_t bytes);
  void *bounce = nullptr;
  size_t bounce_bytes = 0;
  std::vector<char> hbuf;
  void ensure_bounce(size_t bytes);
  void ensure_hbuf(size_t bytes);
};

void comm_unique_id(char out[128]);
int comm_rccl_selftest();   // size-1 communicator on the current device (tests)

}  // namespace lmp_le
