"""Per-bag orchestration shared by the two stage CLIs (the roles of the reference's pipeline_stage1_extract.py
= S1 and pipeline_stage2_filter.py = S2).  Bags are independent, so besides the reference's strictly serial
walk (the default) the children can be fanned out over several GPUs: `PCP_GPUS=0,1,2,3` runs one child per
listed device at a time, each with HIP_VISIBLE_DEVICES set to its device.  Messages, skip rules and the
continue-on-error policy follow S1:20-75 / S2:9-50."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from queue import Queue

HERE = os.path.dirname(os.path.abspath(__file__))
S1_EXTRINSICS = (-0.5777, 0.4385, -0.6884, 0.7814,             # S1:13-18, row-major base<-camera
                 0.7992, 0.1327, -0.5862, 0.4271,
                 -0.1657, -0.8889, -0.4271, 0.5267,
                 0.0, 0.0, 0.0, 1.0)


def gpu_list():
    spec = os.environ.get("PCP_GPUS", "").strip()
    return [g for g in spec.split(",") if g != ""] or [None]


class Children:
    """Runs `python <script> args...` children the way S1:20-30 / S2:9-18 do (PYTHONPATH dropped, failure
    reported and tolerated), on a pool of device slots."""

    def __init__(self, devices=None):
        self.devices = devices or gpu_list()
        self.slots = Queue()
        for d in self.devices:
            self.slots.put(d)

    def run(self, argv):
        dev = self.slots.get()
        try:
            env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
            if dev is not None:
                env["HIP_VISIBLE_DEVICES"] = str(dev)
            print(f"\nExecuting: {sys.executable} {' '.join(argv)}", flush=True)
            ok = subprocess.run([sys.executable] + list(argv), text=True, env=env).returncode == 0
            if not ok:
                print(f"Error executing command: {' '.join(argv)}", flush=True)
            return ok
        finally:
            self.slots.put(dev)

    def map(self, jobs):
        """jobs: callables taking this runner; serial with one slot, else one thread per slot."""
        if len(self.devices) == 1:
            return [job(self) for job in jobs]
        with ThreadPoolExecutor(max_workers=len(self.devices)) as pool:
            return list(pool.map(lambda job: job(self), jobs))


def bag_directories(bag_dir):
    """Sub-directories of bag_dir that hold a rosbag2 storage file (S1:44-45)."""
    out = []
    for name in os.listdir(bag_dir):
        path = os.path.join(bag_dir, name)
        if os.path.isdir(path) and any(f.endswith((".db3", ".mcap")) for f in os.listdir(path)):
            out.append(path)
    return out


def stage1(bag_dir, output_root, extrinsics_csv, runner=None):
    """Topic extraction (the reference's CPU script, if present) then GPU point-cloud extraction per bag."""
    runner = runner or Children()
    os.makedirs(output_root, exist_ok=True)
    bags = bag_directories(bag_dir)
    if not bags:
        print(f"No valid bags found in {bag_dir}")
        return []
    topics = os.environ.get("PCP_EXTRACT_TOPICS_SCRIPT", os.path.join(HERE, "extract_topics_one_camera.py"))
    clouds = os.path.join(HERE, "extract_point_clouds_one_cam.py")

    def one(bag_path):
        def job(r):
            bag = os.path.basename(bag_path.rstrip("/"))
            print(f"\n=== PROCESSING BAG: {bag} ===", flush=True)
            extracted_root = os.path.join(output_root, bag, "extracted_data")
            os.makedirs(extracted_root, exist_ok=True)
            actual = os.path.join(extracted_root, bag)                 # the extractor nests <bag> once more (S1:67)
            if os.path.exists(topics):
                if not r.run([topics, "--bag_path", bag_path, "--output_root", extracted_root]):
                    return False
            elif not os.path.isdir(os.path.join(actual, "camera", "rgb")):
                print(f"extract_topics_one_camera.py not available and {actual} has no extracted data, skipping...")
                return False
            pcd_root = os.path.join(output_root, bag, "point_clouds")
            os.makedirs(pcd_root, exist_ok=True)
            if not r.run([clouds, "--bag_path", bag_path, "--extracted_data_root", actual, "--output_root", pcd_root,
                          f"--extrinsics={extrinsics_csv}"]):
                return False
            print(f"Stage 1 complete for: {bag}", flush=True)
            return True
        return job

    return runner.map([one(b) for b in bags])


def stage2(output_root, runner=None):
    """Crop + FPS of every bag's global-frame clouds into <bag>/filtered_point_clouds."""
    runner = runner or Children()
    if not os.path.exists(output_root):
        print(f"Directory not found: {output_root}")
        return []
    script = os.path.join(HERE, "dp3_point_cloud_processing.py")

    def one(bag):
        def job(r):
            print(f"\n=== FILTERING PCD: {bag} ===", flush=True)
            src = os.path.join(output_root, bag, "point_clouds", "global_frame")
            if not os.path.exists(src):
                print(f"global_frame not found for {bag}, skipping...")       # S2:40-42
                return False
            if not r.run([script, "--input_root", src, "--output_root",
                          os.path.join(output_root, bag, "filtered_point_clouds")]):
                return False
            print(f"Stage 2 complete for: {bag}", flush=True)
            return True
        return job

    bags = [d for d in os.listdir(output_root) if os.path.isdir(os.path.join(output_root, d))]
    return runner.map([one(b) for b in bags])
