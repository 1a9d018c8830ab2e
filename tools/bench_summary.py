import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    sd=d.get("second_dataset",{})
    print(f.split("/")[-1], "val",d["value"],"res",d["resident"]["value"],"sus",d.get("sustained",{}).get("value"),"trav",r["avg_launch_ms"],"replay",r["replay_avg_launch_ms"],"rerank",r["rerank_avg_launch_ms"],"solo",r.get("solo_avg_launch_ms"),"| clustered",sd.get("value"),sd.get("sustained"),"|",d["cpu_baseline"]["gpu_matches_oracle_on_sample"] if d.get("cpu_baseline") else None,d.get("lean_replays"))
