// shared between the CPU-test emulation units (parse_emu.cc, pipeline_emu.cc)
#pragma once
#include <string>
#include <vector>
#include "batch_layout.h"

struct EmuBatch {
  hipdec::BatchLayout L;
  std::vector<uint8_t> arena;
  int32_t status = 0;
  std::string err;
};

// a sequence of pictures (the samples of a track): the decoder instance's state between them - POC state and decoded picture buffer - plus the
// emulated batches whose memory the reference pictures live in
struct EmuSeq {
  hipdec::SeqContext ctx;
  std::vector<EmuBatch*> alive;                       // every batch decoded so far (reference planes point into their arenas / full frames)
  std::vector<std::vector<uint8_t>*> full_frames;     // uncropped post-SAO copies made for pictures with a conformance window
};
