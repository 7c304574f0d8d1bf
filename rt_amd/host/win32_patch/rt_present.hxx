// rt_present.hxx — UNTESTED Windows half of SURVEY.md §8f rows 1 and 4 (see README.md).
// Included by rt.hpp when the module lives in the reference tree (code/rt/).
#pragma once

namespace rt {
// The traced image as a texture ImGui can show, and the state the control panel edits.
struct Rt_Present {
  ITexture2D                 *texture;        // d3d_layer.hxx aliases
  ::ID3D11ShaderResourceView *view;
  u32    width, height;
  u32   *rgba;                 // width*height resolved frame, alloc_perm (base/memory.cxx:23-34)
  Camera          camera;
  Render_Settings settings;    // spp = samples traced per displayed frame
  Vec3   cam_from, cam_at;
  f32    vfov_deg, aperture, focus_dist;
  bool   restart;              // the next frame starts the accumulation again
  f32    mrays_per_s;          // readout: rays of the last frame / its trace-kernel time
  u32    samples_shown;
};
extern Rt_Present gRt_Present;

void rt_present_init_or_panic(u32 width, u32 height);
void rt_present_shutdown();
// One iteration of the frame loop: trace settings.spp more samples (rt_render_resolved_or_panic:
// the float sums stay on the GPU), then rt_present_upload.
void rt_present_frame_or_panic();
// rgba8 (0xRRGGBBAA, row 0 = bottom) -> the texture (RGBA bytes, row 0 = top)
void rt_present_upload(u32 const *rgba8);
// Inside dear_imgui_update(), between ImGui::NewFrame() and the end of the function: the image
// window and the control panel (replaces ShowDemoWindow, imgui.cpp:25-26).
void rt_present_panel();
}  // namespace rt
