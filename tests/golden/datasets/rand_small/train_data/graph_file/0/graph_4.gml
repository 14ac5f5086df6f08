graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  edge [
    source 0
    target 1
    capacity 8883
  ]
  edge [
    source 0
    target 8
    capacity 5391
  ]
  edge [
    source 1
    target 0
    capacity 8883
  ]
  edge [
    source 1
    target 2
    capacity 2861
  ]
  edge [
    source 2
    target 1
    capacity 2861
  ]
  edge [
    source 2
    target 3
    capacity 9311
  ]
  edge [
    source 2
    target 4
    capacity 1993
  ]
  edge [
    source 3
    target 2
    capacity 9311
  ]
  edge [
    source 3
    target 4
    capacity 2904
  ]
  edge [
    source 4
    target 2
    capacity 1993
  ]
  edge [
    source 4
    target 3
    capacity 2904
  ]
  edge [
    source 4
    target 5
    capacity 8935
  ]
  edge [
    source 5
    target 4
    capacity 8935
  ]
  edge [
    source 5
    target 6
    capacity 8405
  ]
  edge [
    source 6
    target 5
    capacity 8405
  ]
  edge [
    source 6
    target 7
    capacity 4438
  ]
  edge [
    source 7
    target 6
    capacity 4438
  ]
  edge [
    source 7
    target 8
    capacity 2107
  ]
  edge [
    source 8
    target 0
    capacity 5391
  ]
  edge [
    source 8
    target 7
    capacity 2107
  ]
]
