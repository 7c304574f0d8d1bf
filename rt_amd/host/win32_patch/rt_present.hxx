// rt_present.hxx — UNTESTED Windows half of SURVEY.md §8f rows 1 and 4 (see README.md).
// Included by rt.hpp when the module lives in the reference tree (code/rt/).
#pragma once

namespace rt {
// The traced image as a texture ImGui can show, and the state the control panel edits.
struct Rt_Present {
  ITexture2D                 *texture;        // d3d_layer.hxx aliases
  ::ID3D11ShaderResourceView *view;
  u32   *rgba;                 // width*height resolved frame, alloc_perm (base/memory.cxx:23-34)
  Rt_Panel_State panel;        // rt.hpp: camera, settings, restart flag, readouts - the platform-free
                               // state machine that rt_headless --script runs on Linux (tests/test_headless.py)
};
extern Rt_Present gRt_Present;

void rt_present_init_or_panic(u32 width, u32 height);
void rt_present_shutdown();
// One iteration of the frame loop: rt_panel_step_or_panic (settings.spp more samples, the float sums
// stay on the GPU), then rt_present_upload.
void rt_present_frame_or_panic();
// rgba8 (0xRRGGBBAA, row 0 = bottom) -> the texture (RGBA bytes, row 0 = top)
void rt_present_upload(u32 const *rgba8);
// Inside dear_imgui_update(), between ImGui::NewFrame() and the end of the function: the image
// window and the control panel (replaces ShowDemoWindow, imgui.cpp:25-26).
void rt_present_panel();
}  // namespace rt
