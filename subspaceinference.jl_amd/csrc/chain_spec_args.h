// Kernel arguments of the shape-specialised persistent RWMH loop (chain_spec.inc: si_spec_grid_kernel).  Plain C types only:
// this header is compiled by hipcc into the library AND handed to hiprtc with the kernel's source at run time.
#pragma once
struct SiSpecGridArgs {
  const double *swa, *P, *X, *Y;
  const int* perm;                 // natural weight index -> fragment order (si_spec_perm_kernel)
  double *wbuf, *ybuf;             // per chain: 4 weight vectors in fragment order, [parity][accepted, rejected] (w_stride apart), and the model outputs (y_stride apart); handed between workgroups
  long long w_stride, y_stride;
  unsigned* cnt;                   // 8 x 32 words (eight 128-byte lines: the shards of the barrier counter) per chain, zeroed before the launch
  unsigned* status;                // raised by a workgroup whose barrier wait timed out
  double *Z_out, *lp_out;
  long long* nacc_out;
  long long ldP, itr;
  unsigned long long seed;
  double sigma_z, c0, sigma2;
  int N, M, G, B, chain_id0, nblocks;
  int y_in_lds, o_y, o_blk, o_z, o_red, o_flag;   // LDS offsets (doubles) behind the images of the forward
};
