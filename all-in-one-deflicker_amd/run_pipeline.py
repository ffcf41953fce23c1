"""Drop-in for the reference's end-to-end driver `test.py` (test.py:1-43): same flags, same three stages, same folders,
run from the root of a checkout of the reference — but stage 1 is this package's MI355X path and `--gpu` is actually
forwarded (the reference parses it and drops it, test.py:36-42).

    python <this repo>/all-in-one-deflicker_amd/run_pipeline.py --video_name data/test/X.mp4 [--fps 10] [--gpu 0] [--class_name C]

Stage 0 (ffmpeg frame extraction) and stage 2 (`src/neural_filter_and_refinement.py`) are the reference's own commands,
unchanged; the flow / mask preprocessors are called by the stage-1 CLI exactly as the reference's stage-1 scripts do."""
import argparse
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_commands(opts):
    """The shell commands of the three stages, in order (pure function: unit-tested without running anything)."""
    cmds = []
    if opts.video_name is not None:
        base = os.path.basename(opts.video_name)[:-4]
        folder = "./data/test/{}".format(base)
        cmds.append(("mkdir", folder))
        cmds.append(("sh", "ffmpeg -i {} -vf fps={} -start_number 0 {}/%05d.png".format(opts.video_name, opts.fps, folder)))
    else:
        base = os.path.basename(opts.video_frame_folder)
        folder = "./data/test/{}".format(base)
        if not os.path.isdir(folder):
            cmds.append(("sh", "mv {} {}".format(base, folder)))
    py = sys.executable or "python"
    if opts.class_name is None:
        cmds.append(("sh", "{} {} --vid_name {} --gpu {}".format(py, os.path.join(_HERE, "stage1.py"), base, opts.gpu)))
    else:
        cmds.append(("sh", "{} {} --vid_name {} --class_name {} --gpu {}".format(py, os.path.join(_HERE, "stage1_seg.py"), base, opts.class_name, opts.gpu)))
    cmds.append(("sh", "python src/neural_filter_and_refinement.py --video_name {} --fps {}".format(base, opts.fps)))
    return cmds


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt_filter", default="./pretrained_weights/neural_filter.pth", type=str)
    p.add_argument("--ckpt_local", default="./pretrained_weights/local_refinement_net.pth", type=str)
    p.add_argument("--video_name", default=None, type=str)
    p.add_argument("--video_frame_folder", default=None, type=str)
    p.add_argument("--fps", default=10, type=int)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--class_name", default=None, type=str)
    opts = p.parse_args(argv)
    if opts.video_name is None and opts.video_frame_folder is None:
        p.error("--video_name or --video_frame_folder")
    print(opts)
    for kind, c in build_commands(opts):
        print(c)
        if kind == "mkdir":
            os.makedirs(c, exist_ok=True)
        elif os.system(c) != 0:
            sys.exit("command failed: " + c)


if __name__ == "__main__":
    main()
