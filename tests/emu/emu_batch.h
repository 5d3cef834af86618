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
