// graphmat/mpi_single.h -- single-process stand-in for the handful of MPI calls
// GraphMat applications make in their main() (MPI_Init/Finalize/Barrier, rank and
// size queries).  Used only when no real <mpi.h> is on the include path or when
// GRAPHMAT_NO_MPI is defined.  The engine itself never calls MPI: multi-GPU runs
// are one process per GPU with the exchange done over RCCL.
#ifndef GRAPHMAT_MPI_SINGLE_H_
#define GRAPHMAT_MPI_SINGLE_H_
#include <sys/time.h>
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_INT 1
#define MPI_LAND 1
#define MPI_MAX 2
#define MPI_SUM 3
static inline int MPI_Init(int*, char***) { return MPI_SUCCESS; }
static inline int MPI_Finalize(void) { return MPI_SUCCESS; }
static inline int MPI_Barrier(MPI_Comm) { return MPI_SUCCESS; }
static inline int MPI_Comm_rank(MPI_Comm, int* r) { *r = 0; return MPI_SUCCESS; }
static inline int MPI_Comm_size(MPI_Comm, int* n) { *n = 1; return MPI_SUCCESS; }
static inline double MPI_Wtime(void) {
  struct timeval tv;
  gettimeofday(&tv, 0);
  return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}
#endif
