"""Where the per-iteration cost of the pipelined frame sits: wave-time profile (stats build, unfused launches) of the 4K frame
at two workspace sizes.  avg wave lifetime / launch duration = how much of a launch the average wave is resident."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_FUSE"] = "0"
os.environ["RT_LANES"] = "1"
from raytracer_amd import FlatScene
from raytracer_amd.renderer import Renderer
W, H, SPP = 3840, 2160, 64
for gb in sys.argv[1:] or ("8", "96"):
    os.environ["RT_WORKSPACE_GB"] = gb
    r = Renderer(0).set_scene(FlatScene.builtin("competition"))
    r.render(W, H, spp=SPP, want=())
    r.render(W, H, spp=SPP, want=())
    t = r.render(W, H, spp=SPP, collect_stats=2, want=())["stats"]          # plain kernels, per-class times
    st = r.render(W, H, spp=SPP, collect_stats=5 | 2, want=())["stats"]     # profile build: clocks + per-class times of ITS launches
    print(f"== workspace {gb} GB: {st['chunks']} chunks; plain frame {t['ms_total']:.1f} ms: closest {t['ms_closest']:.1f} any {t['ms_any']:.1f} "
          f"shade {t['ms_shade']:.1f} packet {t['ms_packet']:.1f}; profile build frame {st['ms_total']:.1f} ms")
    for name, ms, launches in (("prof_closest", st["ms_closest"], st["launches_closest"]), ("prof_any", st["ms_any"], st["launches_any"])):
        p = st[name]
        life = p[7]
        nw = 2048 * 4  # waves of a traversal launch (8 blocks per CU)
        per_wave_launch = life / (nw * max(launches, 1))
        print(f"  {name}: {launches} launches, {ms:.1f} ms; wave lifetime sum {life:.3e} clocks = {per_wave_launch:.0f} per wave and launch; "
              f"launch {ms / max(launches,1) * 1e3:.0f} us -> {per_wave_launch / (ms / max(launches,1) * 1e3):.1f} clocks per us of launch")
        print("    refill %.1f%%  node %.1f%%  tri %.1f%%  other %.1f%% of wave lifetime; refills %d (%.0f clocks each, %.1f rays each); node phases %.3e (%.0f clocks, %.1f lanes) tri %.3e (%.0f clocks, %.1f lanes)"
              % (100 * p[0] / life, 100 * p[1] / life, 100 * p[4] / life, 100 * (life - p[0] - p[1] - p[4]) / life, p[8], p[0] / max(p[8], 1), p[9] / max(p[8], 1),
                 p[2], p[1] / max(p[2], 1), p[3] / max(p[2], 1), p[5], p[4] / max(p[5], 1), p[6] / max(p[5], 1)))
    p = st["prof_shade"]
    tot = sum(p[:5])
    print(f"  prof_shade: iterations {p[5]}, clocks/iteration {tot / max(p[5], 1):.0f}, shade ms {st['ms_shade']:.1f}; sections % " + " ".join(f"{100 * p[k] / tot:.1f}" for k in range(5)))
    r.close()
