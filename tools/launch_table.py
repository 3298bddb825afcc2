#!/usr/bin/env python
"""ONE scenario of the launch table (tools/launch_table.sh runs each under `rocprofv3 --kernel-trace`): a warm-up pass, a marker
(an `hilc_tail` launch of 7 elements: its kernel name separates warm-up from the counted pass in the trace), then ONE counted pass
through the product's modules.  Prints the entry points of the counted pass (`ops.timed_launches()` records: kind + tag).
  python tools/launch_table.py <scenario>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hilcodec_amd
from hilcodec_amd import ops, synth

SCENARIOS = {
    # name: (model, mode, batch, samples per clip / hop, exec options)
    "offline hil_speech B256 T24000": ("hil_speech", "offline", 256, 24000, {}),
    "offline hil_music B256 T24000": ("hil_music", "offline", 256, 24000, {}),
    "offline ragged T5000 B7": ("hil_speech", "offline", 7, 5000, {}),
    "offline T4802 (T % 4 != 0) B3": ("hil_speech", "offline", 3, 4802, {}),
    "offline B640 T24000 (largest activation >= 4 GiB: clip chunks)": ("hil_speech", "offline", 640, 24000, {}),
    "offline, stage_launches off": ("hil_speech", "offline", 64, 24000, {"stage_launches": False}),
    "offline, stage_launches off, wide_blocks off": ("hil_speech", "offline", 64, 24000, {"stage_launches": False, "wide_blocks": False}),
    "streaming hop 320, 1024 streams": ("hil_speech", "streaming", 1024, 320, {}),
    "streaming hop 320, 1024 streams, hil_music": ("hil_music", "streaming", 1024, 320, {}),
    "streaming hop 320, 37 streams (ragged runs)": ("hil_speech", "streaming", 37, 320, {}),
    "streaming long hop 1280 (4 frames), 64 streams": ("hil_speech", "streaming", 64, 1280, {}),
    "streaming hop 320, decoder_stage_narrow off (PipelinedHop's capture)": ("hil_speech", "streaming", 1024, 320, {"decoder_stage_narrow": False}),
    "streaming hop 320, stage_launches off": ("hil_speech", "streaming", 256, 320, {"stage_launches": False}),
    "streaming hop 320, stage_launches off, wide_blocks off": ("hil_speech", "streaming", 256, 320, {"stage_launches": False, "wide_blocks": False}),
}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "--list":
        print("\n".join(SCENARIOS))
        sys.exit(0)
    name = sys.argv[1]
    model_name, mode, B, T, opts = SCENARIOS[name]
    dev = torch.device("cuda:0")
    mk = synth.model_kwargs(model_name)
    sd = synth.synth_state_dict(model_name, 7)
    nq = mk["vq_kwargs"]["num_quantizers"]
    if mode == "offline":
        model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
        model.load_state_dict(sd, strict=False)
        for l in model.quantizer.layers:
            l.initted = True
        x = synth.synth_clips(B, T).to(dev)

        def step():
            z = model.encoder(x)
            q, _, _, idx = model.quantizer(z, None, return_indices=True)
            return model.decoder(q)
    else:
        from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
        smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
        model = StreamingHILCodec(24000, **smk).eval()
        model.load_offline_state_dict(sd)
        model.remove_weight_reparameterizations()
        x = synth.synth_clips(B, T, seed=4321).to(dev)
        state = list(model.initialize_cache(x))

        def step():
            z, state[0] = model.encoder(x, *state[0])
            idx = model.quantizer(z, nq)
            q = model.dequantizer(idx, nq)
            wav, state[1] = model.decoder(q, *state[1])
            return wav
    for half in (model.encoder, model.decoder):
        for k, v in opts.items():
            setattr(half.exec_options, k, v)
    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        marker = torch.zeros(1, 1, 7, device=dev)
        ops.tail(marker, None, 3)                     # hist_out / tail kernel on 7 elements = the marker in the kernel trace
        torch.cuda.synchronize()
        with ops.timed_launches() as t:
            step()
        torch.cuda.synchronize()
    print(f"## {name}: {len(t.records)} launches")
    for kind, work, e0, e1, tag in t.records:
        print(f"   {kind:12s} {tag}")
