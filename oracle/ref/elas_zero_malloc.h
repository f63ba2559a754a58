// Forced in front of the reference's libelas sources (oracle/ref/Makefile: -include).  The sources read memory they
// allocated and never wrote: Elas::adaptiveMean's scratch image D_tmp outside the rows / columns its horizontal pass
// covers (CPU/elas.cpp:1361-1369 initialise only the invalid pixels; the vertical pass reads column 3 and rows 0-2 and
// the last three), and the descriptor images' borders (descriptor.cpp:31, filled for v in [3, height-3), u in [3, width-3)
// only; findMatch reads row 2 / row height-3 and columns 2 / width-3).  In a fresh process those bytes are zero pages;
// in a long-running one whatever the heap holds.  This header pins them to zero so that the compiled reference is a
// function of its inputs: malloc -> calloc, _mm_malloc -> a zeroing one.  Nothing else of the sources changes.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#include <cstdlib>
#include <emmintrin.h>
#include <pmmintrin.h>

static inline void* ref_zeroing_mm_malloc(size_t size, size_t align) {
  void* p = _mm_malloc(size, align);
  if (p) memset(p, 0, size);
  return p;
}
#define malloc(x) calloc(1, (x))
#define _mm_malloc(s, a) ref_zeroing_mm_malloc((s), (a))
