namespace gato { template __global__ void pcgc_dual2_kernel<Indy7>(Buffers, int, int, uint32_t, int, float, int); }
