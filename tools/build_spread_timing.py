"""Build torch-pme_amd/libmipme_timing.so: libmipme with clock stamps in the spread kernel (for tools/spread_phases.py).

    python tools/build_spread_timing.py          # here (hipcc cross-compiles), then on the GPU box:
    MIPME_LIB=$PWD/torch-pme_amd/libmipme_timing.so python tools/spread_phases.py

Copies csrc/ to a scratch directory, patches bricks.hip (thread 0 of every spread workgroup writes s_memrealtime, 100 MHz,
at the phase boundaries into a __device__ array; adds mipme_debug_spread_times to read it back) and builds the variant
library next to the real one.  The product library never carries the stamps.
"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="mipme_timing_")
os.makedirs(os.path.join(tmp, "torch-pme_amd"))
shutil.copytree(os.path.join(ROOT, "torch-pme_amd", "csrc"), os.path.join(tmp, "torch-pme_amd", "csrc"),
                ignore=shutil.ignore_patterns("*.o"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
SRC = os.path.join(tmp, "torch-pme_amd", "csrc", "bricks.hip")
s = open(SRC).read()
def rep(a,b):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b)
rep("static constexpr int SPREAD_THREADS = 512;",'''__device__ long long g_spread_t[8 * 8192];
#define STAMP(k) do { if (threadIdx.x == 0 && block < 8192) g_spread_t[block * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
static constexpr int SPREAD_THREADS = 512;''')
rep('''  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (clear_count && threadIdx.x == 0) clear_count[block] = 0;''','''  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  STAMP(0);
  if (clear_count && threadIdx.x == 0) clear_count[block] = 0;''')
rep('''  const int total = maxlen;  // longest of the 27 candidate lists
''','''  const int total = maxlen;  // longest of the 27 candidate lists
  STAMP(1);
''')
rep('''      __syncthreads();
      const int ns = nsurv;''','''      __syncthreads();
      STAMP(2);
      const int ns = nsurv;''')
rep('''dst[BRICK + 1 + k] = wr[k];
        }
        __syncthreads();''','''dst[BRICK + 1 + k] = wr[k];
        }
        __syncthreads();
        STAMP(3);''')
rep('''        }
        __syncthreads();
      }
    }
    // R: sum''','''        }
        STAMP(6);
        __syncthreads();
        STAMP(4);
      }
    }
    // R: sum''')
rep('''      if (gx < g.nx && gy < g.ny && gz < g.nz) mesh[c * M + gx * plane + int64_t(gy) * g.nz + gz] = v;
    }
    __syncthreads();
  }
}''','''      if (gx < g.nx && gy < g.ny && gz < g.nz) mesh[c * M + gx * plane + int64_t(gy) * g.nz + gz] = v;
    }
    __syncthreads();
    STAMP(5);
  }
}''')
s=s.rstrip()+'''

extern "C" int mipme_debug_spread_times(void* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mipme::g_spread_t), size_t(n) * 8);
}
'''
open(SRC, "w").write(s)
subprocess.check_call(["make", "-C", os.path.dirname(SRC), "TARGET=" + os.path.join(ROOT, "torch-pme_amd", "libmipme_timing.so")])
shutil.rmtree(tmp)
