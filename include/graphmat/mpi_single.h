// graphmat/mpi_single.h -- stand-in for the handful of MPI calls GraphMat applications make in their
// main() (MPI_Init/Finalize/Barrier, rank and size queries).  Used when no real <mpi.h> is on the include
// path or when GRAPHMAT_NO_MPI is defined.
//
// Started as one plain process these calls do nothing (rank 0 of 1).  Started once per GPU by a launcher that
// sets rank and size in the environment (torch.distributed.run / torchrun, mpirun, srun, or by hand:
// GRAPHMAT_RANK, GRAPHMAT_NRANKS, GRAPHMAT_LOCAL_RANK), MPI_Init joins the library's RCCL communicator
// (graphmat_hip.h: gm_dist_init_from_env; the unique id travels through a rendezvous file), and Graph<V,E>
// then builds this rank's shard of a 1-D row-sharded graph (Graph.h), exchanging messages natively over RCCL.
// The engine itself never calls MPI.
#ifndef GRAPHMAT_MPI_SINGLE_H_
#define GRAPHMAT_MPI_SINGLE_H_
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>

#include "../graphmat_hip.h"

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_INT 1
#define MPI_LAND 1
#define MPI_MAX 2
#define MPI_SUM 3
static inline int MPI_Init(int*, char***) {
  int r = 0, n = 1;
  if (gm_dist_init_from_env(&r, &n) != GM_OK) {
    printf("GraphMat(HIP): MPI_Init: %s\n", gm_last_error());
    exit(1);
  }
  return MPI_SUCCESS;
}
static inline int MPI_Finalize(void) {
  gm_dist_finalize();
  return MPI_SUCCESS;
}
static inline int MPI_Barrier(MPI_Comm) {
  gm_dist_barrier();
  return MPI_SUCCESS;
}
static inline int MPI_Comm_rank(MPI_Comm, int* r) {
  int n = 0;
  gm_dist_info(r, &n);
  if (n == 0) *r = 0;
  return MPI_SUCCESS;
}
static inline int MPI_Comm_size(MPI_Comm, int* n) {
  int r = 0;
  gm_dist_info(&r, n);
  if (*n == 0) *n = 1;
  return MPI_SUCCESS;
}
static inline double MPI_Wtime(void) {
  struct timeval tv;
  gettimeofday(&tv, 0);
  return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}
#endif
