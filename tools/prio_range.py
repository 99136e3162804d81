import torch
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for p in (-2, -1, 0, 1):
    try:
        s = torch.cuda.Stream(priority=p)
        print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "failed:", e)
