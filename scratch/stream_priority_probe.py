

def main():
    import torch
    print("priority_range", torch.cuda.Stream.priority_range())
    for p in (1, 0, -1):
        try:
            s = torch.cuda.Stream(priority=p)
            print("priority", p, "->", s.priority)
        except Exception as e:
            print("priority", p, "failed:", e)


if __name__ == "__main__":
    main()
