"""Wall-clock accumulator with the interface of the reference's ``smart_crossover/timer.py``:
start_timer / end_timer add one interval to ``total_duration``; ``accumulate_time`` adds a
solver-reported duration.  ``Output.runtime`` of the network crossover is defined through it."""
import datetime


class Timer:
    def __init__(self) -> None:
        self.clear()

    def clear(self) -> None:
        self.start = datetime.datetime.min
        self.end = datetime.datetime.min
        self.total_duration = datetime.timedelta(0)

    def start_timer(self) -> None:
        self.start = datetime.datetime.now()

    def end_timer(self) -> None:
        self.end = datetime.datetime.now()
        self.total_duration += self.end - self.start

    def accumulate_time(self, new_duration: datetime.timedelta) -> None:
        self.total_duration += new_duration
